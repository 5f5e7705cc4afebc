// Probe (GPU box, round 6): weight gradient of the level-0 transposed convolution (dW[ci 64][4 phases x 32] = sum over
// 32 x 128 x 128 pixels of x[p][ci] * d_up[p][col]) with BOTH operands loaded straight into the MFMA operand registers:
// the contraction index of v_mfma_f32_16x16x4_f32 is the PIXEL (4 per instruction), lane (t16, g) loads 16 bytes = four
// consecutive channels 4 t16 .. + 3 of pixel p0 + g (one instruction = four whole 256-byte pixel rows), and MFMA m of a
// step takes element m of that load as its A operand (rows = channels 4 i + m).  No LDS, no barrier in the loop; a wave
// keeps the whole 64 x 128 block of dW in 128 accumulator registers.  Timing only (the cross-wave sum and the slab
// stores of the real kernel are replaced by one store per wave).  Compared with wgrad_dma_kernel<1>: 140 us cold.
//   hipcc -w --offload-arch=gfx950 -O3 tools/probes/pw_wgrad_probe.hip -o build/exp/pw_wgrad_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int K = 64, CO = 32;
constexpr int SETS = 4;

template <int DEPTH, int WAVES_PER_SIMD>
__global__ __launch_bounds__(256 * WAVES_PER_SIMD, 1) void pw_wgrad(const float* __restrict__ x, const float* __restrict__ up,
                                                                    float* __restrict__ out, int NI, int H, int W) {
  const int tid = threadIdx.x, lane = tid & 63, t16 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), n_waves = blockDim.x >> 6;
  const long rows = static_cast<long>(NI) * H, total_w = static_cast<long>(gridDim.x) * n_waves;
  const long wi = static_cast<long>(blockIdx.x) * n_waves + wave;
  const long r0 = rows * wi / total_w, r1 = rows * (wi + 1) / total_w;
  const int steps_per_row = W / 4;
  const long n_steps = (r1 - r0) * steps_per_row;

  f32x4 acc[4][2][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[m][h][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 dbs[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

  // per-lane constants: x: channel quad; d_up: output row phase a = load index, column phase b and channel quad from t16
  const int ca = 4 * t16;
  const int cb = (t16 >> 3) * CO + 4 * (t16 & 7);  // pixel 2 x + (t16 >> 3) of the up row, channels 4 (t16 & 7) ..
  f32x4 A[DEPTH], B[DEPTH][2];
  long s_issue = 0;
  auto issue = [&](int slot) {
    const long s = s_issue++;
    const long row = r0 + s / steps_per_row;
    const int x0 = static_cast<int>(s % steps_per_row) * 4 + g;
    A[slot] = *reinterpret_cast<const f32x4*>(x + (row * W + x0) * K + ca);
#pragma unroll
    for (int a = 0; a < 2; ++a)
      B[slot][a] = *reinterpret_cast<const f32x4*>(up + ((2 * row + a) * (2 * W) + 2 * x0) * CO + cb);
  };
#pragma unroll
  for (int i = 0; i < DEPTH; ++i)
    if (i < n_steps) issue(i);
  for (long s = 0; s < n_steps; s += DEPTH) {
#pragma unroll
    for (int slot = 0; slot < DEPTH; ++slot) {
      if (s + slot < n_steps) {
        const f32x4 av = A[slot];
        const f32x4 b0 = B[slot][0], b1 = B[slot][1];
        if (s + slot + DEPTH < n_steps) issue(slot);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dbs[0][e] += b0[e];
          dbs[1][e] += b1[e];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[m][0][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], b0[q], acc[m][0][q], 0, 0, 0);
            acc[m][1][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], b1[q], acc[m][1][q], 0, 0, 0);
          }
      }
    }
  }
  f32x4 t = dbs[0] + dbs[1];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < 4; ++q) t += acc[m][h][q];
  *reinterpret_cast<f32x4*>(out + (wi * 64 + lane) * 4) = t;
}

template <int DEPTH, int WPS>
void bench(const float* dx, const float* dup, float* dout, int NI, int H, int W, int wgs) {
  const long P = static_cast<long>(NI) * H * W;
  int it = 0;
  auto run = [&] {
    const int s = (it++) % SETS;
    hipLaunchKernelGGL((pw_wgrad<DEPTH, WPS>), dim3(wgs), dim3(256 * WPS), 0, 0, dx + s * P * K, dup + s * P * 4 * CO, dout, NI, H, W);
  };
  for (int i = 0; i < 20; ++i) run();
  hipDeviceSynchronize();
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipEventRecord(e0);
  for (int i = 0; i < 100; ++i) run();
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double us = ms * 1e3 / 100, bytes = static_cast<double>(P) * (K + 4 * CO) * 4, flop = 2.0 * P * K * 4 * CO;
  printf("pw_wgrad depth %d, %d waves/SIMD, %d workgroups: %.1f us  %.2f TB/s  %.1f TFLOP/s\n", DEPTH, WPS, wgs, us, bytes / us * 1e-6,
         flop / us * 1e-6);
}

int main() {
  const int NI = 32, H = 128, W = 128;
  const long P = static_cast<long>(NI) * H * W;
  float *dx, *dup, *dout;
  hipMalloc(&dx, SETS * P * K * 4);
  hipMalloc(&dup, SETS * P * 4 * CO * 4);
  hipMalloc(&dout, 4096 * 64 * 16);
  hipMemset(dx, 0, SETS * P * K * 4);
  hipMemset(dup, 0, SETS * P * 4 * CO * 4);
  std::vector<float> h(1 << 20);
  for (auto& v : h) v = (rand() % 2001 - 1000) * 1e-3f;
  for (long o = 0; o + (1 << 20) <= SETS * P * K; o += 1 << 20) hipMemcpy(dx + o, h.data(), 4 << 20, hipMemcpyHostToDevice);
  for (long o = 0; o + (1 << 20) <= SETS * P * 4 * CO; o += 1 << 20) hipMemcpy(dup + o, h.data(), 4 << 20, hipMemcpyHostToDevice);
  for (int rep = 0; rep < 2; ++rep) {
    bench<2, 2>(dx, dup, dout, NI, H, W, 256);
    bench<3, 2>(dx, dup, dout, NI, H, W, 256);
    bench<4, 2>(dx, dup, dout, NI, H, W, 256);
    bench<6, 2>(dx, dup, dout, NI, H, W, 256);
    bench<4, 1>(dx, dup, dout, NI, H, W, 256);
    bench<8, 1>(dx, dup, dout, NI, H, W, 256);
    bench<4, 1>(dx, dup, dout, NI, H, W, 512);
  }
  return 0;
}

// Probe (GPU box, round 6): the fp32 pointwise GEMM of the level-0 transposed convolution (K = 64 input channels,
// N = 4 phases x 32 output channels, 32 x 128 x 128 low-resolution pixels) WITHOUT staging the activations in LDS:
// every lane loads its own B-operand fragments (16 bytes = 4 channels of one pixel) straight from global memory, the
// weights sit in LDS for the kernel's lifetime as the A operand (rows = output columns), and the accumulator of
// v_mfma_f32_16x16x4_f32 then holds four consecutive output channels of the lane's pixel: 16-byte stores from registers.
// No barrier in the steady state, four waves per SIMD.  Compared with gemm_fast_kernel<1> (149 us on this shape).
//   hipcc -w --offload-arch=gfx950 -O3 tools/probes/pw_direct_probe.hip -o build/exp/pw_direct_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int K = 64, CO = 32, N = 4 * CO;  // level-0 shape
constexpr int NCB = N / 16, NQ = K / 16;
constexpr int SETS = 4;  // operand sets the timed launches rotate over: 1.6 GB, nothing survives in the 256 MB Infinity Cache

// weight image: [q][cb][g][t16][e] = W[k = 16 q + 4 g + e][col = 16 cb + t16]
// PB = 16-pixel blocks per wave tile; PRE: the next tile's loads are issued before this tile's MFMAs (second fragment set);
// NTS: non-temporal stores
template <int WGS, int PB, int PRE, int NTS>
__global__ __launch_bounds__(256, WGS) void pw_direct_fwd(const float* __restrict__ x, const float* __restrict__ wimg,
                                                           const float* __restrict__ bias, float* __restrict__ out, int NI,
                                                           int H, int W) {
  __shared__ __attribute__((aligned(16))) float w_lds[K * N];
  __shared__ __attribute__((aligned(16))) float b_lds[N];
  const int tid = threadIdx.x, lane = tid & 63, t16 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  for (int i = tid; i < K * N / 4; i += 256) reinterpret_cast<f32x4*>(w_lds)[i] = reinterpret_cast<const f32x4*>(wimg)[i];
  if (tid < N) b_lds[tid] = bias[tid];
  __syncthreads();

  const int tiles_per_row = W / (16 * PB);
  const long n_tiles = static_cast<long>(NI) * H * tiles_per_row;
  const long stride = static_cast<long>(gridDim.x) * 4;
  long t = static_cast<long>(blockIdx.x) * 4 + wave;
  if (t >= n_tiles) return;

  f32x4 X[PB][NQ], XN[PB][NQ];
  auto issue_loads = [&](long tile, f32x4 (&X)[PB][NQ]) {
    const long row = tile / tiles_per_row;  // n * H + y
    const int x0 = static_cast<int>(tile - row * tiles_per_row) * 16 * PB;
    const float* base = x + (row * W + x0 + t16) * K + 4 * g;
#pragma unroll
    for (int pb = 0; pb < PB; ++pb)
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (NTS == 4) X[pb][q] = f32x4{static_cast<float>(tile), 1.f, 2.f, static_cast<float>(q)};
        else X[pb][q] = *reinterpret_cast<const f32x4*>(base + (16 * pb) * K + 16 * q);
      }
  };
  issue_loads(t, X);
  const float* wl = w_lds + lane * 4;
  while (true) {
    const long tn = t + stride;
    const bool more = tn < n_tiles;
    if (PRE && more) issue_loads(tn, XN);
    f32x4 acc[PB][NCB];
#pragma unroll
    for (int cb = 0; cb < NCB; ++cb) {
      const f32x4 b4 = *reinterpret_cast<const f32x4*>(&b_lds[16 * cb + 4 * g]);
#pragma unroll
      for (int pb = 0; pb < PB; ++pb) acc[pb][cb] = b4;
    }
    // weight fragments: hand-placed reads one step ahead (asm volatile: hipcc would hoist all 32 loop-invariant reads out
    // of the tile loop -- 128 registers -- and spill)
    const unsigned wa = static_cast<unsigned>(reinterpret_cast<uintptr_t>(wl));
    f32x4 wf[2];
    asm volatile("ds_read_b128 %0, %1" : "=v"(wf[0]) : "v"(wa) : "memory");
#pragma unroll
    for (int i = 0; i < NQ * NCB; ++i) {
      const int q = i / NCB, cb = i % NCB;
      if (i + 1 < NQ * NCB) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[(i + 1) & 1]) : "v"(wa), "n"((i + 1) * 1024) : "memory");
      if (i + 1 < NQ * NCB) asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wf[i & 1]));
      else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[i & 1]));
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
          if (NTS != 3 || (i & 7) == 0) acc[pb][cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i & 1][e], X[pb][q][e], acc[pb][cb], 0, 0, 0);
    }
    // this tile's geometry for the stores, then the next tile's loads (the fragment registers are free), then the stores
    const long row = t / tiles_per_row;
    const int n = static_cast<int>(row / H), y = static_cast<int>(row - static_cast<long>(n) * H);
    const int x0 = static_cast<int>(t - row * tiles_per_row) * 16 * PB;
    if (!PRE && more) issue_loads(tn, X);
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
      const int xx = x0 + 16 * pb + t16;
#pragma unroll
      for (int cb = 0; cb < NCB; ++cb) {
        const int ph = cb >> 1, a = ph >> 1, b = ph & 1;
        float* dst = out + ((static_cast<long>(n) * 2 * H + 2 * y + a) * (2 * W) + 2 * xx + b) * CO + 16 * (cb & 1) + 4 * g;
        if (NTS == 5 || NTS == 6) {  // timing experiment (wrong placement of the values): one store = eight pixels x 128 contiguous bytes
          const int xs = x0 + 16 * pb + (t16 & 7) + 8 * (cb & 1);
          dst = out + ((static_cast<long>(n) * 2 * H + 2 * y + a) * (2 * W) + 2 * xs + b) * CO + 16 * (t16 >> 3) + 4 * g;
        }
        if (NTS == 2) {
          if (acc[pb][cb][0] == 123.456f) *reinterpret_cast<f32x4*>(dst) = acc[pb][cb];
        } else if (NTS == 1 || NTS == 6) __builtin_nontemporal_store(acc[pb][cb], reinterpret_cast<f32x4*>(dst));
        else *reinterpret_cast<f32x4*>(dst) = acc[pb][cb];
      }
    }
    if (!more) break;
    t = tn;
    if (PRE) {
#pragma unroll
      for (int pb = 0; pb < PB; ++pb)
#pragma unroll
        for (int q = 0; q < NQ; ++q) X[pb][q] = XN[pb][q];
    }
  }
}

template <int WGS, int PB, int PRE, int NTS>
void bench(const float* dx, const float* dw, const float* db, float* dout, int NI, int H, int W) {
  const long P = static_cast<long>(NI) * H * W;
  int it = 0;
  auto run = [&] {
    const int s = (it++) % SETS;
    hipLaunchKernelGGL((pw_direct_fwd<WGS, PB, PRE, NTS>), dim3(256 * WGS), dim3(256), 0, 0, dx + s * P * K, dw, db, dout + s * P * N, NI, H, W);
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
  const double us = ms * 1e3 / 100, bytes = static_cast<double>(P) * (K + N) * 4, flop = 2.0 * P * K * N;
  printf("pw_direct_fwd WGS %d PB %d PRE %d NTS %d: %.1f us  %.2f TB/s  %.1f TFLOP/s\n", WGS, PB, PRE, NTS, us, bytes / us * 1e-6, flop / us * 1e-6);
}

int main(int argc, char** argv) {
  const int NI = 32, H = 128, W = 128;
  const long P = static_cast<long>(NI) * H * W;
  std::vector<float> hx(P * K), hw(K * N), hb(N), himg(K * N);
  srand(1);
  for (auto& v : hx) v = (rand() % 2001 - 1000) * 1e-3f;
  for (auto& v : hw) v = (rand() % 2001 - 1000) * 5e-5f;
  for (auto& v : hb) v = (rand() % 2001 - 1000) * 1e-3f;
  for (int q = 0; q < NQ; ++q)
    for (int cb = 0; cb < NCB; ++cb)
      for (int g = 0; g < 4; ++g)
        for (int t16 = 0; t16 < 16; ++t16)
          for (int e = 0; e < 4; ++e)
            himg[((((q * NCB + cb) * 4 + g) * 16 + t16) * 4) + e] = hw[(16 * q + 4 * g + e) * N + 16 * cb + t16];
  float *dx, *dw, *db, *dout;
  hipMalloc(&dx, SETS * P * K * 4);
  hipMalloc(&dw, K * N * 4);
  hipMalloc(&db, N * 4);
  hipMalloc(&dout, SETS * P * N * 4);
  for (int s = 0; s < SETS; ++s) hipMemcpy(dx + s * P * K, hx.data(), P * K * 4, hipMemcpyHostToDevice);
  hipMemcpy(dw, himg.data(), K * N * 4, hipMemcpyHostToDevice);
  hipMemcpy(db, hb.data(), N * 4, hipMemcpyHostToDevice);
  hipMemset(dout, 0, SETS * P * N * 4);
  for (int rep = 0; rep < 2; ++rep) {
    bench<4, 1, 0, 0>(dx, dw, db, dout, NI, H, W);
    bench<4, 1, 0, 5>(dx, dw, db, dout, NI, H, W);  // full-line stores
    bench<4, 1, 0, 6>(dx, dw, db, dout, NI, H, W);  // full-line non-temporal stores
    bench<4, 1, 0, 1>(dx, dw, db, dout, NI, H, W);
    bench<4, 1, 0, 4>(dx, dw, db, dout, NI, H, W);  // no loads
  }
  bench<4, 1, 0, 0>(dx, dw, db, dout, NI, H, W);
  std::vector<float> ho(P * N);
  hipMemcpy(ho.data(), dout, P * N * 4, hipMemcpyDeviceToHost);
  double worst = 0;
  for (int s = 0; s < 4000; ++s) {
    const long p = (static_cast<long>(rand()) * 7919 + s) % P;
    const int col = rand() % N;
    const int n = p / (H * W), y = (p / W) % H, xx = p % W;
    double ref = hb[col];
    for (int k = 0; k < K; ++k) ref += static_cast<double>(hx[p * K + k]) * hw[k * N + col];
    const int ph = col / CO, a = ph >> 1, b = ph & 1, co = col % CO;
    const float got = ho[((static_cast<long>(n) * 2 * H + 2 * y + a) * (2 * W) + 2 * xx + b) * CO + co];
    worst = fmax(worst, fabs(got - ref));
  }
  printf("max abs error on 4000 samples: %.3g\n", worst);
  return 0;
}

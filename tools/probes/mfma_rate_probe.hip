// MFMA issue-rate probe (GPU box): how much of the fp32 matrix pipe do W waves per SIMD reach with NACC independent
// v_mfma_f32_16x16x4_f32 accumulators each, operands constant or rewritten by VALU between batches?
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_rate_probe.hip -o build/exp/mfma_rate_probe && build/exp/mfma_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int VALU_PER_BATCH, int MODE>
__global__ __launch_bounds__(512) void k(float* out, int iters, float seed) {
  f32x4 acc[NACC];
#pragma unroll
  for (int i = 0; i < NACC; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a[16], b[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    a[i] = seed + threadIdx.x * 0.001f + i;
    b[i] = seed - threadIdx.x * 0.002f + i;
  }
  // MODE 1: the second wave of each SIMD starts half a batch late; 2 / 3: it runs at wave priority 1 / 3
  if (MODE == 1 && threadIdx.x >= 256) __builtin_amdgcn_s_sleep(4);
  if (MODE == 2 && threadIdx.x >= 256) __builtin_amdgcn_s_setprio(1);
  if (MODE == 3 && threadIdx.x >= 256) __builtin_amdgcn_s_setprio(3);
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int v = 0; v < VALU_PER_BATCH; ++v) a[v & 15] = a[v & 15] * 1.0001f + b[(v + 1) & 15];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i & 15], b[i & 15], acc[i], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC, int VALU, int MODE = 0>
void run(int waves_per_simd, float* out) {
  const int iters = 4000, threads = 64 * 4 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NACC, VALU, MODE>), dim3(256), dim3(threads), 0, 0, out, 10, 1.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NACC, VALU, MODE>), dim3(256), dim3(threads), 0, 0, out, iters, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = 256.0 * 4 * waves_per_simd * iters * NACC * 2048.0;
  printf("accumulators %2d, VALU per batch %3d, mode %d, waves/SIMD %d: %7.1f TFLOP/s (%.2f of 157.3)\n", NACC, VALU, MODE, waves_per_simd,
         flop / ms * 1e-9, flop / ms * 1e-9 / 157.3);
}

int main() {
  float* out;
  hipMalloc(&out, 256 * 512 * 4);
  for (int w = 1; w <= 2; ++w) {
    run<16, 0>(w, out);
    run<32, 0>(w, out);
    run<16, 16>(w, out);
    run<16, 48>(w, out);
    run<32, 48>(w, out);
    run<32, 96>(w, out);
  }
  // sustained rate: ten back-to-back launches of ~0.2 s each of the densest variant (does a power / clock governor cap it?)
  for (int rep = 0; rep < 12; ++rep) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 240000;
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<32, 0, 0>), dim3(256), dim3(512), 0, 0, out, iters, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double flop = 256.0 * 8 * iters * 32 * 2048.0;
    printf("sustained launch %2d: %6.1f ms, %7.1f TFLOP/s (%.2f of 157.3)\n", rep, ms, flop / ms * 1e-9, flop / ms * 1e-9 / 157.3);
  }
  run<16, 48, 1>(2, out);
  run<16, 48, 2>(2, out);
  run<16, 48, 3>(2, out);
  run<32, 96, 1>(2, out);
  run<32, 96, 2>(2, out);
  run<32, 96, 3>(2, out);
  return 0;
}

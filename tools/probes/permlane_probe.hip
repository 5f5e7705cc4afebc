// Probe (GPU box): lane semantics of v_permlane32_swap as used by gemm_bf16.hip's register-direct epilogue.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/permlane_probe.hip -o /tmp/permlane_probe && /tmp/permlane_probe
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* o) {
  unsigned a = threadIdx.x, b = 1000 + threadIdx.x;
  auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  o[threadIdx.x] = r[0];
  o[64 + threadIdx.x] = r[1];
}
int main() {
  unsigned* d;
  unsigned h[128];
  hipMalloc(&d, sizeof(h));
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
  printf("r[0]: lane0=%u lane31=%u lane32=%u lane63=%u\n", h[0], h[31], h[32], h[63]);
  printf("r[1]: lane0=%u lane31=%u lane32=%u lane63=%u\n", h[64], h[95], h[96], h[127]);
  return 0;
}

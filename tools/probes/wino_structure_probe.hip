// Structure probe for the fp32 Winograd GEMM (GPU box, round 6): what does the fp32 matrix pipe lose to
//   (1) one filler instruction of each class placed between / among v_mfma_f32_16x16x4_f32, with one or two waves per SIMD;
//   (2) the synchronisation structure of a (chunk = 64 MFMAs + a non-matrix "load" segment) stream: free running, two
//       independent 4-wave workgroups per CU with a barrier per chunk (the round-1..5 kernel), one 8-wave workgroup in
//       lockstep, one 8-wave workgroup with waves 4-7 half a chunk behind (two barriers per chunk), strict ping-pong.
//   hipcc -w --offload-arch=gfx950 -O3 tools/probes/wino_structure_probe.hip -o build/exp/wino_structure_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

enum Filler { F_NONE, F_FMA, F_PKADD, F_MOV, F_IADD, F_DSREAD, F_DSWRITE, F_SALU, F_READLANE, F_DSREAD2, F_MOV64, F_PKMOV, F_DSREAD128, F_DSREAD2ST64, F_DSWRITE64, F_ADD, F_CNDMASK, F_DPPADD, F_DSWRITE128 };
static const char* kFillerName[] = {"none", "v_fma_f32", "v_pk_add_f32", "v_mov_b32", "v_add_u32", "ds_read_b32", "ds_write_b32",
                                    "s_add_u32", "v_readlane", "ds_read2_b32", "v_mov_b64", "v_pk_mov_b32", "ds_read_b128", "ds_read2st64_b64", "ds_write_b64", "v_add_f32", "v_cndmask_b32", "v_add_f32 dpp", "ds_write_b128"};

template <int KIND>
__device__ __forceinline__ void filler(float (&a)[16], int i, unsigned lds_addr, unsigned& sacc, int& iacc) {
  if constexpr (KIND == F_FMA) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i & 7]) : "v"(a[8 + (i & 3)]), "v"(a[12 + (i & 3)]));
  if constexpr (KIND == F_PKADD) {
    f32x2 p = {a[(2 * i) & 7], a[(2 * i + 1) & 7]}, q = {a[8], a[9]};
    asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(p) : "v"(q));
    a[(2 * i) & 7] = p[0];
    a[(2 * i + 1) & 7] = p[1];
  }
  if constexpr (KIND == F_MOV) asm volatile("v_mov_b32 %0, %1" : "=v"(a[i & 7]) : "v"(a[8 + (i & 3)]));
  if constexpr (KIND == F_IADD) asm volatile("v_add_u32 %0, %0, %1" : "+v"(iacc) : "v"(i + 1));
  if constexpr (KIND == F_DSREAD) asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(a[i & 7]) : "v"(lds_addr), "n"(256 * 0) : "memory");
  if constexpr (KIND == F_DSREAD2) {
    f32x2 p;
    asm volatile("ds_read2_b32 %0, %1 offset0:0 offset1:10" : "=v"(p) : "v"(lds_addr) : "memory");
    a[(2 * i) & 7] = p[0];
    a[(2 * i + 1) & 7] = p[1];
  }
  if constexpr (KIND == F_MOV64) {
    f32x2 p, q = {a[8], a[9]};
    asm volatile("v_mov_b64 %0, %1" : "=v"(p) : "v"(q));
    a[(2 * i) & 7] = p[0];
    a[(2 * i + 1) & 7] = p[1];
  }
  if constexpr (KIND == F_PKMOV) {
    f32x2 p, q = {a[8], a[9]};
    asm volatile("v_pk_mov_b32 %0, %1, %1 op_sel:[0,1]" : "=v"(p) : "v"(q));
    a[(2 * i) & 7] = p[0];
    a[(2 * i + 1) & 7] = p[1];
  }
  if constexpr (KIND == F_DSREAD128) {
    f32x4 p;
    asm volatile("ds_read_b128 %0, %1" : "=v"(p) : "v"(lds_addr & ~15u) : "memory");
    a[i & 3] = p[0];
    a[4 + (i & 3)] = p[3];
  }
  if constexpr (KIND == F_DSREAD2ST64) {
    f32x4 p;
    asm volatile("ds_read2st64_b64 %0, %1 offset0:0 offset1:1" : "=v"(p) : "v"(lds_addr & ~7u) : "memory");
    a[i & 3] = p[0];
    a[4 + (i & 3)] = p[3];
  }
  if constexpr (KIND == F_DSWRITE64) {
    f32x2 q = {a[8], a[9]};
    asm volatile("ds_write_b64 %0, %1" ::"v"(lds_addr & ~7u), "v"(q) : "memory");
  }
  if constexpr (KIND == F_DSWRITE128) {
    f32x4 q = {a[8], a[9], a[10], a[11]};
    asm volatile("ds_write_b128 %0, %1" ::"v"(lds_addr & ~15u), "v"(q) : "memory");
  }
  if constexpr (KIND == F_ADD) asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i & 7]) : "v"(a[8 + (i & 3)]));
  if constexpr (KIND == F_CNDMASK) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a[i & 7]) : "v"(a[8 + (i & 3)]) : );
  if constexpr (KIND == F_DPPADD) asm volatile("v_add_f32_dpp %0, %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf" : "+v"(a[i & 7]) : "v"(a[8 + (i & 3)]));
  if constexpr (KIND == F_DSWRITE) asm volatile("ds_write_b32 %0, %1" ::"v"(lds_addr), "v"(a[8 + (i & 3)]) : "memory");
  if constexpr (KIND == F_SALU) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sacc));
  if constexpr (KIND == F_READLANE) {
    unsigned r;
    asm volatile("v_readlane_b32 %0, %1, 3" : "=s"(r) : "v"(a[8 + (i & 3)]));
    sacc += r;
  }
}

// PLACE 0: NF fillers in a burst between batches of 16 MFMAs; PLACE 1: NF / 16 fillers behind every MFMA
template <int KIND, int NF, int PLACE, int LSTRIDE>
__global__ __launch_bounds__(512) void k_fill(float* out, int iters, float seed) {
  __shared__ float lds[4096];
  f32x4 acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a[16], b[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    a[i] = seed + threadIdx.x * 0.001f + i;
    b[i] = seed - threadIdx.x * 0.002f + i;
  }
  lds[threadIdx.x] = seed;
  __syncthreads();
  const unsigned lds_addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>(&lds[((threadIdx.x & 63) * LSTRIDE) & 2047]));
  unsigned sacc = 0;
  int iacc = 0;
  for (int it = 0; it < iters; ++it) {
    if constexpr (PLACE == 0) {
#pragma unroll
      for (int v = 0; v < NF; ++v) filler<KIND>(a, v, lds_addr, sacc, iacc);
      if constexpr (KIND == F_DSREAD || KIND == F_DSREAD2 || KIND == F_DSWRITE || KIND >= F_DSREAD128 && KIND <= F_DSWRITE64 || KIND == F_DSWRITE128) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[8 + (i & 7)], b[i], acc[i], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    } else {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[8 + (i & 7)], b[i], acc[i], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int v = 0; v < NF / 16; ++v) filler<KIND>(a, i * (NF / 16) + v, lds_addr, sacc, iacc);
        __builtin_amdgcn_sched_barrier(0);
      }
      if constexpr (KIND == F_DSREAD || KIND == F_DSREAD2 || KIND == F_DSWRITE || KIND >= F_DSREAD128 && KIND <= F_DSWRITE64 || KIND == F_DSWRITE128) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
  float s = static_cast<float>(sacc) + static_cast<float>(iacc);
#pragma unroll
  for (int i = 0; i < 16; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3] + a[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static float* g_out;
template <int KIND, int NF, int PLACE, int LSTRIDE = 1>
void run_fill(int waves_per_simd) {
  const int iters = 3000, threads = 256 * waves_per_simd;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k_fill<KIND, NF, PLACE, LSTRIDE>), dim3(256), dim3(threads), 0, 0, g_out, 10, 1.f);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_fill<KIND, NF, PLACE, LSTRIDE>), dim3(256), dim3(threads), 0, 0, g_out, iters, 1.f);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = 256.0 * 4 * waves_per_simd * iters * 16 * 2048.0;
  const double frac = flop / ms * 1e-9 / 157.3;
  // pipe cycles lost per filler: per batch the pipe serves (16 MFMAs x 32 cycles) x waves; time per batch = that / frac
  const double lost = NF > 0 ? (512.0 * waves_per_simd / frac - 512.0 * waves_per_simd) / (NF * waves_per_simd) : 0.0;
  printf("fill %-16s x%3d %-11s lane stride %d dw, waves/SIMD %d: %.3f of peak, %.2f pipe cycles per filler\n", kFillerName[KIND], NF,
         PLACE ? "interleaved" : "burst", LSTRIDE, waves_per_simd, frac, lost);
}

// ---- (2) structure: a chunk = LV filler VALU + a sleep of LS x 64 cycles (the "load" segment L: staging, waits) and 64
// MFMAs in two halves Ma, Mb (+ TV transform VALU in front of each half).  Every EPI-th chunk the load segment is ELONG
// times longer (the epilogue). ----
struct SArgs {
  float* out;
  int chunks;
  int ls, lv, tv, epi_every, epi_ls, epi_lv;
};
template <int NV>
__device__ __forceinline__ void valu_n(float (&a)[16], int n) {  // n rounded down to a multiple of 8
  for (int v = 0; v < n; v += 8) {
#pragma unroll
    for (int u = 0; u < 8; ++u) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[u]) : "v"(a[8 + (u & 3)]), "v"(a[12 + (u & 3)]));
  }
}
__device__ __forceinline__ void mfma32(f32x4 (&acc)[32], const float (&a)[16], const float (&b)[16], int half) {
#pragma unroll
  for (int i = 0; i < 32; ++i) {
    const int j = (half * 32 + i) & 31;
    acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[8 + (i & 7)], b[i & 15], acc[j], 0, 0, 0);
  }
}
__device__ __forceinline__ void seg_l(float (&a)[16], const SArgs& s, int c) {
  const bool epi = s.epi_every > 0 && (c % s.epi_every) == s.epi_every - 1;
  const int ls = epi ? s.epi_ls : s.ls, lv = epi ? s.epi_lv : s.lv;
  valu_n<0>(a, lv);
  for (int i = 0; i < ls; ++i) __builtin_amdgcn_s_sleep(1);
}
// STRUCT 0: free running (no barrier); 1: barrier per chunk (lockstep inside the workgroup); 2: waves >= half of the
// workgroup run half a chunk behind, two barriers per chunk; 3: strict ping-pong (a half computes a whole chunk while the
// other half is in its load segment, barrier between)
template <int STRUCT>
__global__ __launch_bounds__(512) void k_struct(const SArgs s) {
  f32x4 acc[32];
#pragma unroll
  for (int i = 0; i < 32; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  float a[16], b[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    a[i] = 1.f + threadIdx.x * 0.001f + i;
    b[i] = 1.f - threadIdx.x * 0.002f + i;
  }
  const bool late = threadIdx.x >= blockDim.x / 2;
  auto bar = [] { __builtin_amdgcn_s_barrier(); };
  if constexpr (STRUCT == 0 || STRUCT == 1) {
    for (int c = 0; c < s.chunks; ++c) {
      seg_l(a, s, c);
      valu_n<0>(a, s.tv);
      mfma32(acc, a, b, 0);
      valu_n<0>(a, s.tv);
      mfma32(acc, a, b, 1);
      if constexpr (STRUCT == 1) bar();
    }
  } else if constexpr (STRUCT == 2) {
    if (late) bar();
    for (int c = 0; c < s.chunks; ++c) {
      seg_l(a, s, c);
      valu_n<0>(a, s.tv);
      mfma32(acc, a, b, 0);
      bar();
      valu_n<0>(a, s.tv);
      mfma32(acc, a, b, 1);
      bar();
    }
    if (!late) bar();
  } else {
    if (late) bar();
    for (int c = 0; c < s.chunks; ++c) {
      seg_l(a, s, c);
      bar();
      valu_n<0>(a, s.tv);
      mfma32(acc, a, b, 0);
      valu_n<0>(a, s.tv);
      mfma32(acc, a, b, 1);
      bar();
    }
    if (!late) bar();
  }
  float t = 0.f;
#pragma unroll
  for (int i = 0; i < 32; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  s.out[blockIdx.x * blockDim.x + threadIdx.x] = t + a[0];
}

template <int STRUCT>
void run_struct(const char* name, int threads, int wgs_per_cu, SArgs s) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  SArgs w = s;
  w.chunks = 4;
  hipLaunchKernelGGL((k_struct<STRUCT>), dim3(256 * wgs_per_cu), dim3(threads), 0, 0, w);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k_struct<STRUCT>), dim3(256 * wgs_per_cu), dim3(threads), 0, 0, s);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double flop = 256.0 * wgs_per_cu * (threads / 64) * double(s.chunks) * 64 * 2048.0;
  printf("struct %-34s L = %2d x 64 cyc + %3d VALU, T = %2d VALU, epilogue every %d (%3d x 64 cyc + %3d VALU): %.3f of peak\n", name,
         s.ls, s.lv, s.tv, s.epi_every, s.epi_ls, s.epi_lv, flop / ms * 1e-9 / 157.3);
}

int main(int argc, char** argv) {
  hipMalloc(&g_out, 256 * 2 * 512 * 4);
  const bool part1 = argc < 2 || atoi(argv[1]) & 1, part2 = argc < 2 || atoi(argv[1]) & 2;
  if (part1) {
    for (int w = 1; w <= 2; ++w) {
      run_fill<F_NONE, 0, 0>(w);
      run_fill<F_FMA, 48, 0>(w);
      run_fill<F_FMA, 48, 1>(w);
      run_fill<F_FMA, 16, 1>(w);
      run_fill<F_ADD, 48, 1>(w);
      run_fill<F_PKADD, 48, 0>(w);
      run_fill<F_PKADD, 48, 1>(w);
      run_fill<F_PKADD, 16, 1>(w);
      run_fill<F_MOV, 48, 1>(w);
      run_fill<F_MOV64, 48, 1>(w);
      run_fill<F_PKMOV, 48, 1>(w);
      run_fill<F_IADD, 48, 1>(w);
      run_fill<F_CNDMASK, 48, 1>(w);
      run_fill<F_DPPADD, 48, 1>(w);
      run_fill<F_DSREAD, 48, 1>(w);
      run_fill<F_DSREAD, 48, 1, 5>(w);
      run_fill<F_DSREAD2, 48, 1, 1>(w);
      run_fill<F_DSREAD2, 48, 1, 20>(w);
      run_fill<F_DSREAD128, 48, 1, 4>(w);
      run_fill<F_DSREAD2ST64, 48, 1, 2>(w);
      run_fill<F_DSWRITE, 48, 1, 1>(w);
      run_fill<F_DSWRITE64, 48, 1, 2>(w);
      run_fill<F_DSWRITE128, 48, 1, 4>(w);
      run_fill<F_SALU, 48, 1>(w);
      run_fill<F_SALU, 144, 1>(w);
      run_fill<F_READLANE, 16, 1>(w);
    }
  }
  if (part2) {
    // load-segment sizes: ~700 / ~1400 cycles of waiting, 40 VALU; transform 16 VALU per half; epilogue every 4 / 16 chunks
    const SArgs cases[] = {
        {g_out, 3000, 0, 0, 0, 0, 0, 0},       {g_out, 3000, 0, 0, 16, 0, 0, 0},      {g_out, 3000, 10, 40, 16, 0, 0, 0},
        {g_out, 3000, 20, 40, 16, 0, 0, 0},    {g_out, 3000, 10, 40, 16, 4, 40, 400}, {g_out, 3000, 10, 40, 16, 16, 40, 400},
        {g_out, 3000, 20, 40, 16, 4, 50, 400},
    };
    for (const SArgs& s : cases) {
      run_struct<0>("free, 2 x 4-wave WG per CU", 256, 2, s);
      run_struct<0>("free, 1 x 8-wave WG per CU", 512, 1, s);
      run_struct<1>("barrier/chunk, 2 x 4-wave WG per CU", 256, 2, s);
      run_struct<1>("barrier/chunk, 1 x 8-wave lockstep", 512, 1, s);
      run_struct<2>("8-wave, halves half a chunk apart", 512, 1, s);
      run_struct<3>("8-wave, strict ping-pong", 512, 1, s);
    }
  }
  return 0;
}

// BatchNorm finalize inside the launch that takes the statistics (unetpp_bn_fused, include/unetpp_hip.h).
//
// Every workgroup of a persistent kernel adds the (sum, sum of squares) of all its units per column in LDS, publishes
// them as ONE row of the workspace, and takes a ticket; the workgroup that draws the last ticket adds all rows in a
// fixed order (fp64) and writes mean / invstd / scale / shift and the running statistics -- the arithmetic of
// bn_finalize_kernel (pointwise.hip), without its launch (8-13 us per BatchNorm layer, two of them inside the X_0,0
// block).  Hand-off (cdna_hip_programming.md, Guideline 16; results must not depend on dispatch order or placement):
//   producer  rows stored write-through (8-byte agent-scope atomic stores = global_store_dwordx2 sc1), every storing
//             wave drains vmcnt, workgroup barrier, ONE lane adds to the ticket (agent-scope atomic)
//   consumer  (the last arriver, told by the value its add returned) one agent-scope acquire, drained, barrier, then
//             the rows are read with 8-byte agent-scope atomic loads (sc1: never served from this CU's L1)
// The coefficients are consumed by LATER launches of the stream, so they need no hand-off of their own.  The ticket
// word is zero before the launch (the caller allocates it zeroed) and the last arriver leaves it zero.
#pragma once
#include "common.h"

namespace unetpp {

constexpr int kBnFusedMaxCols = 256;   // columns the in-kernel finalize handles (LDS row of the workgroup)
constexpr int kBnFusedRows = 2048;     // >= the largest persistent grid that publishes rows (4 workgroups x 256 CUs)

typedef __attribute__((address_space(1))) unsigned long long gu64_t;
typedef __attribute__((address_space(1))) unsigned gu32_t;

// run: LDS [Ncols][2] sums of this workgroup (complete and visible: the caller has passed a barrier since the last
// update); scratch: LDS, >= (THREADS / 32) * 32 * 2 doubles, 8-byte aligned; flag: one LDS word.
// Must be reached by ALL threads of EVERY workgroup of the grid, exactly once.
template <int THREADS>
__device__ __forceinline__ void bn_fused_finish(const unetpp_bn_fused& bn, float* rows, int Ncols, const float* run,
                                                double* scratch, unsigned* flag) {
  constexpr int SLICES = THREADS / 32;
  const int tid = threadIdx.x;
  gu64_t* grow = (gu64_t*)(rows) + static_cast<long>(blockIdx.x) * Ncols;
  for (int c = tid; c < Ncols; c += THREADS) {
    const unsigned long long v = static_cast<unsigned long long>(__builtin_bit_cast(unsigned, run[2 * c])) |
                                 (static_cast<unsigned long long>(__builtin_bit_cast(unsigned, run[2 * c + 1])) << 32);
    __hip_atomic_store(grow + c, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its row stores ...
  __syncthreads();                                  // ... before the one lane signals for all of them
  if (tid == 0) {
    const unsigned t = __hip_atomic_fetch_add((gu32_t*)bn.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = (t == gridDim.x - 1) ? 1u : 0u;
    if (t == gridDim.x - 1) {
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  if (*flag == 0u) return;
  // ---- the last arriver: all rows are published ----
  const int col32 = tid & 31, slice = tid >> 5;
  const gu64_t* all = (const gu64_t*)(rows);
  const int n_rows = static_cast<int>(gridDim.x);
  for (int c0 = 0; c0 < Ncols; c0 += 32) {
    const int c = c0 + col32;
    double s1 = 0.0, s2 = 0.0;
    if (c < Ncols) {
      for (int r = slice; r < n_rows; r += SLICES) {
        const unsigned long long v = __hip_atomic_load(all + static_cast<long>(r) * Ncols + c, __ATOMIC_RELAXED,
                                                       __HIP_MEMORY_SCOPE_AGENT);
        s1 += static_cast<double>(__builtin_bit_cast(float, static_cast<unsigned>(v)));
        s2 += static_cast<double>(__builtin_bit_cast(float, static_cast<unsigned>(v >> 32)));
      }
    }
    scratch[(slice * 32 + col32) * 2 + 0] = s1;
    scratch[(slice * 32 + col32) * 2 + 1] = s2;
    __syncthreads();
    if (slice == 0 && c < Ncols) {
      s1 = 0.0;
      s2 = 0.0;
      for (int k = 0; k < SLICES; ++k) {  // fixed order
        s1 += scratch[(k * 32 + col32) * 2 + 0];
        s2 += scratch[(k * 32 + col32) * 2 + 1];
      }
      const double cnt = static_cast<double>(bn.count);
      const double m = s1 / cnt;
      double var = s2 / cnt - m * m;
      if (var < 0.0) var = 0.0;
      const double is = 1.0 / sqrt(var + static_cast<double>(bn.eps));
      const double sc = static_cast<double>(bn.gamma[c]) * is;
      bn.mean[c] = static_cast<float>(m);
      bn.invstd[c] = static_cast<float>(is);
      bn.scale[c] = static_cast<float>(sc);
      bn.shift[c] = static_cast<float>(static_cast<double>(bn.beta[c]) - m * sc);
      if (bn.running_mean != nullptr) {
        const double unbiased = bn.count > 1 ? var * cnt / static_cast<double>(bn.count - 1) : var;
        bn.running_mean[c] = static_cast<float>((1.0 - bn.momentum) * bn.running_mean[c] + bn.momentum * m);
        bn.running_var[c] = static_cast<float>((1.0 - bn.momentum) * bn.running_var[c] + bn.momentum * unbiased);
      }
    }
    __syncthreads();
  }
  if (tid == 0) __hip_atomic_store((gu32_t*)bn.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// host side: is the in-kernel finalize possible for this launch?
inline bool bn_fused_in_kernel(const unetpp_gemm_desc* d, int Ncols) {
  return d->bn.scale != nullptr && d->stats_partial != nullptr && Ncols <= kBnFusedMaxCols && d->bn.ticket != nullptr;
}

}  // namespace unetpp

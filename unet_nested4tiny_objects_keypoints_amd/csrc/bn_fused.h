// BatchNorm finalize inside the launch that takes the statistics (unetpp_bn_fused, include/unetpp_hip.h).
//
// Every workgroup of a persistent kernel adds the (sum, sum of squares) of all its units per column in LDS and
// publishes them as ONE row of the workspace.  The rows are summed by the workgroups themselves, in two levels, each
// by a "last arriver": workgroups form groups of 32 (consecutive block indices); the member that draws the group's
// last ticket adds the group's rows (fp64) into one group row, publishes it and takes a ticket of the second level;
// the group that arrives last there adds the <= 64 group rows and writes mean / invstd / scale / shift and the running
// statistics -- the arithmetic of bn_finalize_kernel (pointwise.hip) without its launch (8-13 us per BatchNorm layer,
// two of them inside the X_0,0 block).  Two levels keep the serial tail behind the kernel's last workgroup at two
// batches of <= 32 independent loads per thread (a single last arriver over 1024 rows added 37 us to the 55 us first
// layer); all other group sums run while the rest of the grid still computes.  Fixed summation order: bitwise
// reproducible for a given grid.
//
// Hand-off (cdna_hip_programming.md, Guideline 16; results must not depend on dispatch order or placement):
//   producer  rows stored write-through (8-byte agent-scope atomic stores = global_store_dwordx2 sc1), every storing
//             wave drains vmcnt, workgroup barrier, ONE lane adds to the ticket (agent-scope atomic)
//   consumer  (told by the value its add returned) one agent-scope acquire, drained, barrier, then the rows are read
//             with 8-byte agent-scope atomic loads (sc1: never served from this CU's L1), all requested before any is used
// The coefficients are consumed by LATER launches of the stream, so they need no hand-off of their own.  The ticket
// words are zero before the launch (the caller allocates them zeroed) and the last arrivers leave them zero.
#pragma once
#include "common.h"

namespace unetpp {

constexpr int kBnFusedMaxCols = 256;   // columns the in-kernel finalize handles (one thread per column, LDS row)
constexpr int kBnFusedRows = 2048;     // >= the largest persistent grid that publishes rows (4 workgroups x 256 CUs ..)
constexpr int kBnFusedGroup = 32;      // workgroups per first-level group
constexpr int kBnFusedGroups = kBnFusedRows / kBnFusedGroup;           // 64: second-level rows at most
constexpr int kBnFusedGroupRowsAsRows = kBnFusedGroups * 2;            // fp64 group rows, counted in fp32 rows
constexpr int kBnFusedTickets = 1 + kBnFusedGroups;                    // [0] second level, [1 + g] group g

typedef __attribute__((address_space(1))) unsigned long long gu64_t;
typedef __attribute__((address_space(1))) unsigned gu32_t;

__device__ __forceinline__ void bn_fused_store8(gu64_t* p, unsigned long long v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long bn_fused_load8(const gu64_t* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// every storing wave drains, barrier, one lane takes the ticket; returns (to all threads) whether this workgroup drew
// the last of `expected` tickets, with the acquire done when it did.  flag: one LDS word.
__device__ __forceinline__ bool bn_fused_arrive(unsigned* ticket, unsigned expected, unsigned* flag) {
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // every storing wave drains its row stores ...
  __syncthreads();                                  // ... before the one lane signals for all of them
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add((gu32_t*)ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const bool last = t == expected - 1;
    *flag = last ? 1u : 0u;
    if (last) {
      __hip_atomic_store((gu32_t*)ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // zero for the next launch
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
  }
  __syncthreads();
  const bool last = *flag != 0u;
  __syncthreads();  // the flag word may be reused by the next level
  return last;
}

// run: LDS [Ncols][2] sums of this workgroup (complete and visible: the caller has passed a barrier since the last
// update); rows: the workspace, [gridDim.x] fp32 rows of [Ncols][2] followed (at fp32 row kBnFusedRows or gridDim.x
// rounded up, see group_rows below) by the fp64 group rows; flag: one LDS word.  Ncols <= kBnFusedMaxCols <= THREADS.
// Must be reached by ALL threads of EVERY workgroup of the grid, exactly once.
template <int THREADS>
__device__ __forceinline__ void bn_fused_finish(const unetpp_bn_fused& bn, float* rows, int Ncols, const float* run,
                                                unsigned* flag) {
  static_assert(THREADS >= kBnFusedMaxCols, "one thread per column");
  const int c = threadIdx.x;
  const bool mine = c < Ncols;
  const unsigned n_wg = gridDim.x;
  const unsigned n_groups = (n_wg + kBnFusedGroup - 1) / kBnFusedGroup;
  const unsigned grp = blockIdx.x / kBnFusedGroup;
  const unsigned members = min(static_cast<unsigned>(kBnFusedGroup), n_wg - grp * kBnFusedGroup);
  gu64_t* wg_rows = (gu64_t*)(rows);                                                  // [n_wg][Ncols] (float, float)
  gu64_t* group_rows = wg_rows + static_cast<long>(kBnFusedRows) * Ncols;             // [n_groups][Ncols][2] doubles
  if (mine) {
    const unsigned long long v = static_cast<unsigned long long>(__builtin_bit_cast(unsigned, run[2 * c])) |
                                 (static_cast<unsigned long long>(__builtin_bit_cast(unsigned, run[2 * c + 1])) << 32);
    bn_fused_store8(wg_rows + static_cast<long>(blockIdx.x) * Ncols + c, v);
  }
  if (!bn_fused_arrive(bn.ticket + 1 + grp, members, flag)) return;
  // ---- last arriver of its group: the group's rows -> one fp64 row ----
  {
    unsigned long long v[kBnFusedGroup];
    const gu64_t* base = wg_rows + static_cast<long>(grp) * kBnFusedGroup * Ncols + (mine ? c : 0);
#pragma unroll
    for (int k = 0; k < kBnFusedGroup; ++k)  // all requested before the first is used; rows past the group re-read its last
      v[k] = bn_fused_load8(base + static_cast<long>(min(static_cast<unsigned>(k), members - 1)) * Ncols);
    double s1 = 0.0, s2 = 0.0;
#pragma unroll
    for (int k = 0; k < kBnFusedGroup; ++k) {
      if (static_cast<unsigned>(k) < members) {  // uniform
        s1 += static_cast<double>(__builtin_bit_cast(float, static_cast<unsigned>(v[k])));
        s2 += static_cast<double>(__builtin_bit_cast(float, static_cast<unsigned>(v[k] >> 32)));
      }
    }
    if (mine) {
      bn_fused_store8(group_rows + (static_cast<long>(grp) * Ncols + c) * 2 + 0, __builtin_bit_cast(unsigned long long, s1));
      bn_fused_store8(group_rows + (static_cast<long>(grp) * Ncols + c) * 2 + 1, __builtin_bit_cast(unsigned long long, s2));
    }
  }
  if (!bn_fused_arrive(bn.ticket, n_groups, flag)) return;
  // ---- the last group: all group rows -> coefficients ----
  double s1 = 0.0, s2 = 0.0;
  for (unsigned g0 = 0; g0 < n_groups; g0 += kBnFusedGroup / 2) {
    unsigned long long v[kBnFusedGroup];
    const unsigned left = min(static_cast<unsigned>(kBnFusedGroup / 2), n_groups - g0);
    const gu64_t* base = group_rows + (static_cast<long>(g0) * Ncols + (mine ? c : 0)) * 2;
#pragma unroll
    for (int k = 0; k < kBnFusedGroup / 2; ++k) {
      const long off = static_cast<long>(min(static_cast<unsigned>(k), left - 1)) * Ncols * 2;
      v[2 * k] = bn_fused_load8(base + off);
      v[2 * k + 1] = bn_fused_load8(base + off + 1);
    }
#pragma unroll
    for (int k = 0; k < kBnFusedGroup / 2; ++k) {
      if (static_cast<unsigned>(k) < left) {
        s1 += __builtin_bit_cast(double, v[2 * k]);
        s2 += __builtin_bit_cast(double, v[2 * k + 1]);
      }
    }
  }
  if (!mine) return;
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

// host side: is the in-kernel finalize possible for this launch?
inline bool bn_fused_in_kernel(const unetpp_gemm_desc* d, int Ncols) {
  return d->bn.scale != nullptr && d->stats_partial != nullptr && Ncols <= kBnFusedMaxCols && d->bn.ticket != nullptr;
}

}  // namespace unetpp

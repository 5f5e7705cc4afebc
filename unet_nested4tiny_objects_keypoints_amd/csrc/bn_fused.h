// BatchNorm statistics of a persistent convolution kernel as ONE row per workgroup (unetpp_bn_fused,
// include/unetpp_hip.h): every workgroup adds the (sum, sum of squares) of all its units per column in LDS and writes
// them once at its end, and the library enqueues unetpp_bn_finalize over those <= 2048 rows behind the launch -- instead
// of one row per 256-pixel patch (8192 rows for a level-0 layer of BASELINE configs[1], a 13 us finalize).
//
// Measured and NOT kept (round 3, tools/x00_probe.py): finishing the rows inside the launch, by the last-arriving
// workgroup (agent-scope hand-off: write-through rows, ticket, acquire; cdna_hip_programming.md Guideline 16).  One
// last arriver over the 1024 rows of the first layer added 37 us to its 55 us; two levels (groups of 32 workgroups,
// then the groups) still left 10-15 us behind the last workgroup of each launch -- two memory round trips for the
// ticket, an acquire (buffer_inv, ~1.7 us) and a batch of write-through-coherent loads per level -- i.e. no less than
// the kernel boundary plus the small finalize launch it would replace (~6 us over <= 1024 rows).
#pragma once
#include "common.h"

namespace unetpp {

constexpr int kBnFusedMaxCols = 256;   // columns a workgroup row holds (LDS accumulators of the kernels)
constexpr int kBnFusedRows = 2048;     // >= the largest persistent grid that writes rows (4 workgroups x 256 CUs ..)

// run: LDS [Ncols][2] sums of this workgroup (complete and visible: the caller has passed a barrier since the last
// update) -> row blockIdx.x of rows[gridDim.x][Ncols][2]
template <int THREADS>
__device__ __forceinline__ void bn_rows_store(float* rows, int Ncols, const float* run) {
  float2* dst = reinterpret_cast<float2*>(rows) + static_cast<long>(blockIdx.x) * Ncols;
  for (int c = threadIdx.x; c < Ncols; c += THREADS) dst[c] = float2{run[2 * c], run[2 * c + 1]};
}

// host side: does this launch write per-workgroup rows?
inline bool bn_rows_per_workgroup(const unetpp_gemm_desc* d, int Ncols) {
  return d->bn.scale != nullptr && d->stats_partial != nullptr && Ncols <= kBnFusedMaxCols;
}

}  // namespace unetpp

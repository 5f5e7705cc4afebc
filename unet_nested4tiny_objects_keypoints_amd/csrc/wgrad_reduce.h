// Epilogue shared by the 8-wave weight-gradient kernels (wgrad_fast.hip, wgrad_dma.hip): sum the per-wave
// accumulators of a workgroup and write the workgroup's slab.
#pragma once
#include "common.h"

namespace unetpp {

// Fixed-order tree sum ((w0+w4)+(w2+w6)) + ((w1+w5)+(w3+w7)) through LDS; the total lands in wave 0's registers
// (bitwise reproducible: the order never depends on timing).  rg01 / rg23 each hold two regions of TAPS*1024
// floats; a region is lane-linear [t][r/4][lane][r%4], i.e. 16-byte LDS accesses without bank conflicts and with
// immediate offsets -- the writes and reads of a round issue back to back instead of one read-add-write chain per
// element.  The caller's last tile barrier must have released the staging buffers the regions alias.
template <int TAPS>
__device__ __forceinline__ void tree_sum_waves(f32x16 (&acc)[TAPS], float* rg01, float* rg23, int wave, int lane) {
  constexpr int R = TAPS * 1024;
  auto region = [&](int i) { return ((i & 2) ? rg23 : rg01) + (i & 1) * R + lane * 4; };
  auto put = [&](float* rg) {
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(rg + (t * 4 + q) * 256) =
            f32x4{acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
  };
  auto add = [&](const float* rg) {
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(rg + (t * 4 + q) * 256);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][4 * q + i] += v[i];
      }
  };
  if (wave >= 4) put(region(wave - 4));
  __syncthreads();
  if (wave < 4) add(region(wave));
  __syncthreads();
  if (wave == 2 || wave == 3) put(region(wave - 2));
  __syncthreads();
  if (wave < 2) add(region(wave));
  __syncthreads();
  if (wave == 1) put(region(0));
  __syncthreads();
  if (wave == 0) add(region(0));
}

// Wave 0 stores the summed 32 x 32 block of every tap into the workgroup's slab (MFMA D map: reg r of lane (j, h)
// is row (r&3) + 8*(r>>2) + 4*h, column j).
template <int TAPS>
__device__ __forceinline__ void store_slab_block(const f32x16 (&acc)[TAPS], float* slab, int Ktot, int Ncols, int k0,
                                                 int k_cnt, int n0, int n_cnt, int j, int h) {
  if (j >= n_cnt) return;
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
      if (row < k_cnt) slab[(static_cast<long>(t) * Ktot + k0 + row) * Ncols + n0 + j] = acc[t][r];
    }
}

}  // namespace unetpp

// Unit geometry shared by the persistent multi-view pixel-GEMM kernels (gemm_fast.hip, gemm_wino.hip).
#pragma once
#include "common.h"

namespace unetpp {

struct FastArgs {
  unetpp_gemm_desc d;
  int log2tw, tiles_x, tiles_y;
  int Ktot, Ncols, n_tiles, n_chunks;
  int nt_unit, n_groups;  // column tiles per unit of work, units per pixel patch (n_tiles / nt_unit)
  long total_blocks;
  int bn_in_kernel;  // the BatchNorm partial rows are per workgroup, not per pixel block (bn_fused.h)
};

// 32-bit element offset of a view pixel (the fast kernels only take tensors below 2^31 elements)
__device__ __forceinline__ unsigned view_pixel_offset32(const unetpp_view& v, int n, int y, int x) {
  return ((static_cast<unsigned>(n) * v.Hs + (y * v.sy + v.oy)) * v.Ws + (x * v.sx + v.ox)) * v.C + v.c_off;
}

__device__ __forceinline__ long xcd_remap(long bid, long total) {
  const long q = total >> 3, r = total & 7;
  const long xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// A unit of work = (256-pixel patch, 32-column tile), column tile fastest.  Workgroups are PERSISTENT: the grid is
// at most 3 per CU and every workgroup walks its units as one flat stream of K chunks, so the first chunk of the
// next unit is prefetched under the last MFMA loop of the current one and the epilogue's stores drain under the next
// unit's MFMAs.  (With one unit per workgroup the co-resident workgroups run in lockstep -- all load, all compute,
// all store -- and a short-K unit spends 40 % of its time outside the MFMA loop.)
struct UnitGeom {
  int n, ty0, tx0;  // image, patch origin
  int group;        // first column tile = group * nt_unit
  long patch;       // pixel-patch index (BatchNorm partial row)
};
struct TileCols {
  int nt, ov;     // tile index inside its out view; the out view
  int n0, n_cnt;  // first GEMM column, valid columns
};

template <int LOG2N = 5>
__device__ __forceinline__ TileCols decode_tile(const FastArgs& a, int nt_global) {
  constexpr int NW = 1 << LOG2N;
  int nt = nt_global, ov = 0, col_base = 0;
  while (ov < a.d.n_out - 1) {
    const int tiles_v = (a.d.out[ov].c_len + NW - 1) >> LOG2N;
    if (nt < tiles_v) break;
    nt -= tiles_v;
    col_base += a.d.out[ov].c_len;
    ++ov;
  }
  TileCols t;
  t.nt = nt;
  t.ov = ov;
  t.n0 = col_base + nt * NW;
  t.n_cnt = min(NW, a.d.out[ov].c_len - nt * NW);
  return t;
}

template <int LOG2TW>
__device__ __forceinline__ UnitGeom decode_unit(const FastArgs& a, long lb) {
  constexpr int TW = 1 << LOG2TW, TH = kBlockPixels >> LOG2TW;
  UnitGeom g;
  unsigned bid = static_cast<unsigned>(lb);  // total_blocks < 2^31 (fast_args): 32-bit divisions
  g.group = static_cast<int>(bid % static_cast<unsigned>(a.n_groups));
  bid /= static_cast<unsigned>(a.n_groups);
  g.patch = bid;
  const int txi = static_cast<int>(bid % static_cast<unsigned>(a.tiles_x));
  bid /= static_cast<unsigned>(a.tiles_x);
  const int tyi = static_cast<int>(bid % static_cast<unsigned>(a.tiles_y));
  g.n = static_cast<int>(bid / static_cast<unsigned>(a.tiles_y));
  g.ty0 = tyi * TH;
  g.tx0 = txi * TW;
  return g;
}

// Host side: validates a descriptor for the fast kernels and fills the unit geometry; kc = channels per K chunk,
// ncol = columns per tile.
inline bool fast_args(const unetpp_gemm_desc* d, FastArgs& a, int kc, int ncol = 32) {
  if (d == nullptr || d->N <= 0 || d->H <= 0 || d->W <= 0) return false;
  if (d->taps != 9 && d->taps != 1) return false;
  if (d->n_in < 1 || d->n_in > UNETPP_MAX_VIEWS || d->n_out < 1 || d->n_out > UNETPP_MAX_VIEWS) return false;
  a.d = *d;
  a.bn_in_kernel = 0;
  a.Ktot = a.Ncols = a.n_tiles = a.n_chunks = 0;
  for (int i = 0; i < d->n_in; ++i) {
    const unetpp_view& v = d->in[i];
    if (!view_ok(v) || !view_covers(v, d->H, d->W)) return false;
    if (v.gate != nullptr) return false;  // ReLU gates on load go through the generic kernel
    if (v.scale != nullptr && ((reinterpret_cast<uintptr_t>(v.scale) | reinterpret_cast<uintptr_t>(v.shift)) & 15) != 0)
      return false;
    if (((v.C | v.c_off | v.c_len) & 3) != 0 || (reinterpret_cast<uintptr_t>(v.ptr) & 15) != 0) return false;
    if (static_cast<long>(d->N) * v.Hs * v.Ws * v.C >= 0x7fffffffL) return false;  // 32-bit element offsets
    a.Ktot += v.c_len;
    a.n_chunks += (v.c_len + kc - 1) / kc;
  }
  for (int i = 0; i < d->n_out; ++i) {
    if (!view_ok(d->out[i]) || !view_covers(d->out[i], d->H, d->W)) return false;
    if (static_cast<long>(d->N) * d->out[i].Hs * d->out[i].Ws * d->out[i].C >= 0x7fffffffL) return false;
    a.Ncols += d->out[i].c_len;
    a.n_tiles += (d->out[i].c_len + ncol - 1) / ncol;
  }
  const TileGeom g = tile_geom(d->H, d->W);
  a.log2tw = g.log2tw;
  a.tiles_x = g.tiles_x;
  a.tiles_y = g.tiles_y;
  // column tiles per unit: pointwise GEMMs (deconvolution phases) without a statistics epilogue take 2 (measured:
  // 2 tiles x 3 workgroups per CU beats 4 tiles x 2 and 1 tile x 4 on the deconvolutions)
  a.nt_unit = (d->taps == 1 && d->stats_partial == nullptr && a.n_tiles % 2 == 0) ? 2 : 1;
  a.n_groups = a.n_tiles / a.nt_unit;
  a.total_blocks = static_cast<long>(d->N) * g.tiles_y * g.tiles_x * a.n_groups;
  return a.total_blocks <= 0x7fffffffL;
}


// this workgroup's units: XCD x = blockIdx & 7 owns a contiguous range of units (neighbouring patches share an L2),
// its workgroups take them round-robin.  The launchers make gridDim.x a multiple of 8 whenever a workgroup has more
// than one unit.
struct UnitRange {
  long first, step, count;
};
__device__ __forceinline__ UnitRange my_unit_range(long total_blocks) {
  UnitRange u;
  const long W8 = gridDim.x >> 3;
  if (gridDim.x >= total_blocks) {
    u.first = xcd_remap(blockIdx.x, total_blocks);
    u.step = 0;
    u.count = 1;
  } else {
    const long q = total_blocks >> 3, r = total_blocks & 7;
    const long xcd = blockIdx.x & 7, widx = blockIdx.x >> 3;
    const long start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const long cnt = q + (xcd < r ? 1 : 0);
    u.first = start + widx;
    u.step = W8;
    u.count = widx < cnt ? (cnt - widx + W8 - 1) / W8 : 0;
  }
  return u;
}

// The same XCD ranges, but every workgroup takes a CONTIGUOUS piece (unit order: column tile fastest), so successive
// units of a workgroup mostly share the pixel patch and its geometry can be reused (gemm_wino.hip).
__device__ __forceinline__ UnitRange my_contiguous_unit_range(long total_blocks) {
  UnitRange u;
  const long W8 = gridDim.x >> 3;
  u.step = 1;
  if (gridDim.x >= total_blocks) {
    u.first = xcd_remap(blockIdx.x, total_blocks);
    u.count = 1;
  } else {
    const long q = total_blocks >> 3, r = total_blocks & 7;
    const long xcd = blockIdx.x & 7, widx = blockIdx.x >> 3;
    const long start = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    const long cnt = q + (xcd < r ? 1 : 0);
    const long per = cnt / W8, rem = cnt % W8;
    u.first = start + widx * per + (widx < rem ? widx : rem);
    u.count = per + (widx < rem ? 1 : 0);
  }
  return u;
}

}  // namespace unetpp

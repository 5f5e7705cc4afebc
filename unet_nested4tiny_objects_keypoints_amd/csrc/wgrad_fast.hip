// Fast path of the weight-gradient kernel (same math, MFMA maps and slab format as wgrad.hip) for plain,
// 16-byte aligned views.
//
//   * 512 threads = 8 waves per workgroup, one workgroup per CU (2 waves per SIMD); every wave owns 32 of the
//     tile's 256 pixels and keeps 9 taps x 16 accumulator registers live across the whole pixel loop;
//   * two LDS buffers (x patch with halo 43.5 KB + dy patch 32 KB each, 151 KB in all): while the MFMAs of
//     tile t read buffer t&1, the registers holding tile t+1 are written to the other buffer and the global
//     loads of tile t+2 are issued -- one barrier per tile, HBM/L2 latency and the LDS fill hide under
//     ~9 k cycles of MFMA work;
//   * staging geometry (halo position of every item) is computed once per workgroup.
// The 8 waves are summed through LDS in fixed order; one slab per workgroup (bitwise reproducible).
#include "common.h"

namespace unetpp {
namespace {

constexpr int kWThreads = 512;

struct WFastArgs {
  unetpp_wgrad_desc d;
  int log2tw, tiles_x, tiles_y;
  int Ktot, Ncols, n_tiles_cols;
  long n_pix_tiles;
};

template <int TAPS>
__global__ __launch_bounds__(kWThreads, 2) void wgrad_fast_kernel(const WFastArgs a) {
  constexpr int HALO = (TAPS == 9) ? 1 : 0;
  constexpr int XPIX = (TAPS == 9) ? kMaxHaloPixels : kBlockPixels;
  constexpr int X_FLOATS = XPIX * 32;
  constexpr int DY_FLOATS = kBlockPixels * 32;
  constexpr int BUF = X_FLOATS + DY_FLOATS;
  constexpr int X_ITEMS = (XPIX * 8 + kWThreads - 1) / kWThreads;
  constexpr int DY_ITEMS = (kBlockPixels * 8) / kWThreads;
  extern __shared__ __attribute__((aligned(16))) float smem[];  // 2 * BUF floats

  const unetpp_wgrad_desc& d = a.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;

  int nt = blockIdx.y % a.n_tiles_cols;
  int kt = blockIdx.y / a.n_tiles_cols;
  int dv = 0, col_base = 0;
  while (dv < d.n_dy - 1) {
    const int tiles_v = (d.dy[dv].c_len + 31) >> 5;
    if (nt < tiles_v) break;
    nt -= tiles_v;
    col_base += d.dy[dv].c_len;
    ++dv;
  }
  const unetpp_view& DY = d.dy[dv];
  int xv = 0, kbase = 0;
  while (xv < d.n_x - 1) {
    const int tiles_v = (d.x[xv].c_len + 31) >> 5;
    if (kt < tiles_v) break;
    kt -= tiles_v;
    kbase += d.x[xv].c_len;
    ++xv;
  }
  const unetpp_view& X = d.x[xv];
  const int c0 = kt * 32;
  const int k_cnt = min(32, X.c_len - c0);
  const int nc0 = nt * 32;
  const int n0 = col_base + nc0;
  const int n_cnt = min(32, DY.c_len - nc0);
  const bool want_db = (blockIdx.y / a.n_tiles_cols) == 0;

  const int TW = 1 << a.log2tw, TH = kBlockPixels >> a.log2tw;
  const int HWp = TW + 2 * HALO, HHp = TH + 2 * HALO;
  const int npix = HWp * HHp;

  // ---- staging items, geometry relative to the tile origin (tile-invariant) ----
  int xi_dy[X_ITEMS], xi_dx[X_ITEMS], xi_lds[X_ITEMS];
#pragma unroll
  for (int q = 0; q < X_ITEMS; ++q) {
    const int it = tid + q * kWThreads;
    const int hp = it >> 3, cc = (it & 7) << 2;
    const int hy = hp / HWp;
    xi_dy[q] = hy - HALO;
    xi_dx[q] = hp - hy * HWp - HALO;
    xi_lds[q] = (it < npix * 8 && cc < k_cnt) ? hp * 32 + cc : -1;
  }
  int di_dy[DY_ITEMS], di_dx[DY_ITEMS], di_lds[DY_ITEMS];
#pragma unroll
  for (int q = 0; q < DY_ITEMS; ++q) {
    const int it = tid + q * kWThreads;
    const int p = it >> 3, cc = (it & 7) << 2;
    di_dy[q] = p >> a.log2tw;
    di_dx[q] = p & (TW - 1);
    di_lds[q] = (cc < n_cnt) ? p * 32 + cc : -1;
  }
  // columns / channels that are never staged must read as zero in both buffers
  for (int i = tid; i < 2 * BUF; i += kWThreads) smem[i] = 0.f;
  __syncthreads();

  f32x4 rx[X_ITEMS], rdy[DY_ITEMS];
  auto load_tile = [&](long tile) {
    long b = tile;
    const int txi = static_cast<int>(b % a.tiles_x);
    b /= a.tiles_x;
    const int tyi = static_cast<int>(b % a.tiles_y);
    const int n = static_cast<int>(b / a.tiles_y);
    const int ty0 = tyi * TH, tx0 = txi * TW;
#pragma unroll
    for (int q = 0; q < X_ITEMS; ++q) {
      const int y = ty0 + xi_dy[q], x = tx0 + xi_dx[q];
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (xi_lds[q] >= 0 && y >= 0 && y < d.H && x >= 0 && x < d.W)
        v = *reinterpret_cast<const f32x4*>(X.ptr + view_pixel_offset(X, n, y, x) + c0 + ((xi_lds[q] & 31)));
      rx[q] = v;
    }
#pragma unroll
    for (int q = 0; q < DY_ITEMS; ++q) {
      const int y = ty0 + di_dy[q], x = tx0 + di_dx[q];
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (di_lds[q] >= 0 && y < d.H && x < d.W)
        v = *reinterpret_cast<const f32x4*>(DY.ptr + view_pixel_offset(DY, n, y, x) + nc0 + (di_lds[q] & 31));
      rdy[q] = v;
    }
  };
  auto store_tile = [&](float* buf) {
#pragma unroll
    for (int q = 0; q < X_ITEMS; ++q)
      if (xi_lds[q] >= 0) *reinterpret_cast<f32x4*>(&buf[xi_lds[q]]) = rx[q];
#pragma unroll
    for (int q = 0; q < DY_ITEMS; ++q)
      if (di_lds[q] >= 0) *reinterpret_cast<f32x4*>(&buf[X_FLOATS + di_lds[q]]) = rdy[q];
  };

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float dbsum = 0.f;

  // tiles of this workgroup: blockIdx.x, +gridDim.x, ...
  const long stride = gridDim.x;
  long t0 = blockIdx.x;
  long n_my = (t0 < a.n_pix_tiles) ? (a.n_pix_tiles - t0 + stride - 1) / stride : 0;
  if (n_my > 0) {
    load_tile(t0);
    store_tile(smem);
    if (n_my > 1) load_tile(t0 + stride);
  }
  __syncthreads();
  for (long i = 0; i < n_my; ++i) {
    float* cur = smem + (i & 1) * BUF;
    float* nxt = smem + ((i + 1) & 1) * BUF;
    if (i + 1 < n_my) {
      store_tile(nxt);  // registers hold tile i+1; `nxt` was last read in iteration i-1 (barrier below)
      if (i + 2 < n_my) load_tile(t0 + (i + 2) * stride);
    }
    const float* x_tile = cur;
    const float* dy_tile = cur + X_FLOATS;
#pragma unroll 4
    for (int pp = 0; pp < 16; ++pp) {
      const int p = 32 * wave + 2 * pp + h;
      const float bv = dy_tile[p * 32 + j];
      dbsum += bv;
      const int xb = ((p >> a.log2tw) * HWp + (p & (TW - 1))) * 32 + j;
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int toff = (TAPS == 9) ? ((t / 3) * HWp + (t % 3)) * 32 : 0;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x_tile[xb + toff], bv, acc[t], 0, 0, 0);
      }
    }
    __syncthreads();
  }

  // ---- fixed-order sum of the 8 waves through LDS, then one slab per workgroup ----
  float* red = smem;  // [TAPS][32 k][32 n]
  for (int w = 0; w < 8; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
          const int idx = (t * 32 + row) * 32 + j;
          red[idx] = (w == 0) ? acc[t][r] : red[idx] + acc[t][r];
        }
    }
    __syncthreads();
  }
  const long slab_stride = (static_cast<long>(TAPS) * a.Ktot + 1) * a.Ncols;
  float* slab = d.slabs + blockIdx.x * slab_stride;
  for (int it = tid; it < TAPS * 32 * 32; it += kWThreads) {
    const int col = it & 31, row = (it >> 5) & 31, t = it >> 10;
    if (row < k_cnt && col < n_cnt)
      slab[(static_cast<long>(t) * a.Ktot + kbase + c0 + row) * a.Ncols + n0 + col] = red[it];
  }
  if (want_db) {
    dbsum += __shfl_xor(dbsum, 32);
    float* dbs = smem + TAPS * 1024;
    __syncthreads();
    if (h == 0) dbs[wave * 32 + j] = dbsum;
    __syncthreads();
    if (tid < n_cnt) {
      float s = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) s += dbs[w * 32 + tid];
      slab[static_cast<long>(TAPS) * a.Ktot * a.Ncols + n0 + tid] = s;
    }
  }
}

bool plain_aligned(const unetpp_view& v) {
  return v.scale == nullptr && v.gate == nullptr && !v.relu && ((v.C | v.c_off | v.c_len) & 3) == 0 &&
         (reinterpret_cast<uintptr_t>(v.ptr) & 15) == 0;
}

}  // namespace

// returns UNETPP_OK after launching, or 1 when the descriptor needs the generic kernel
int launch_wgrad_fast(const unetpp_wgrad_desc* d, int Ktot, int Ncols, int n_tiles_cols, int k_tiles, hipStream_t st) {
  for (int i = 0; i < d->n_x; ++i)
    if (!plain_aligned(d->x[i])) return 1;
  for (int i = 0; i < d->n_dy; ++i)
    if (!plain_aligned(d->dy[i])) return 1;
  WFastArgs a;
  a.d = *d;
  a.Ktot = Ktot;
  a.Ncols = Ncols;
  a.n_tiles_cols = n_tiles_cols;
  const TileGeom g = tile_geom(d->H, d->W);
  a.log2tw = g.log2tw;
  a.tiles_x = g.tiles_x;
  a.tiles_y = g.tiles_y;
  a.n_pix_tiles = static_cast<long>(d->N) * g.tiles_y * g.tiles_x;
  const dim3 grid(static_cast<unsigned>(d->n_split), static_cast<unsigned>(static_cast<long>(k_tiles) * n_tiles_cols));
  // > 64 KB of dynamic LDS needs the per-function opt-in; it is idempotent and keeps the ABI stateless
  if (d->taps == 9) {
    constexpr size_t lds = 2 * (kMaxHaloPixels * 32 + kBlockPixels * 32) * sizeof(float);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_fast_kernel<9>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess)
      return UNETPP_ELAUNCH;
    hipLaunchKernelGGL(wgrad_fast_kernel<9>, grid, dim3(kWThreads), lds, st, a);
  } else {
    constexpr size_t lds = 2 * (kBlockPixels * 32 + kBlockPixels * 32) * sizeof(float);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_fast_kernel<1>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess)
      return UNETPP_ELAUNCH;
    hipLaunchKernelGGL(wgrad_fast_kernel<1>, grid, dim3(kWThreads), lds, st, a);
  }
  return launch_status();
}

}  // namespace unetpp

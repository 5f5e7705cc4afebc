// Weight-gradient kernel with LDS-DMA staging (same math, MFMA maps and slab format as wgrad.hip / wgrad_fast.hip)
// for views without any load transform -- all decoder convolutions and deconvolutions, i.e. most of the wgrad work.
//
// The register-staged kernel (wgrad_fast.hip) can only keep a quarter tile of loads in flight (9 taps x 16
// accumulators leave ~12 staging registers), about 1 us of cover, and its waves park at `s_waitcnt vmcnt` for a
// third of their time.  Here the x patch (with halo) and the dy patch of tile t+1 go HBM/L2 -> LDS with
// global_load_lds_dwordx4 -- no staging registers, no ds_write -- while the MFMAs of tile t run: a full tile
// (~9 k MFMA cycles per wave, two waves per SIMD) of cover.
//   * The two tile buffers are two DISTINCT static __shared__ arrays and the tile loop is unrolled by two, so hipcc
//     can prove that the ds_reads of the buffer being computed do not alias the DMA target; with one array and a
//     runtime buffer index it drains vmcnt before every ds_read and the pipeline serialises.
//   * LDS images are the lane-linear [pixel][32 floats] rows the DMA wants (wave-uniform base + lane * 16 B);
//     out-of-image halo pixels are zero-filled by a predicated ds_write of the same item.
//   * 512 threads = 8 waves, one workgroup per CU, one barrier per tile; fixed-order wave reduction; one slab per
//     workgroup (bitwise reproducible).
#include "common.h"
#include "wgrad_reduce.h"

namespace unetpp {
namespace {

constexpr int kWThreads = 512;

struct WDmaArgs {
  unetpp_wgrad_desc d;
  int tiles_x, tiles_y;
  int Ktot, Ncols, n_tiles_cols;
  long n_pix_tiles;
};

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int TAPS, int LOG2TW>
__global__ __launch_bounds__(kWThreads, 2) void wgrad_dma_kernel(const WDmaArgs a) {
  constexpr int HALO = (TAPS == 9) ? 1 : 0;
  constexpr int TW = 1 << LOG2TW, TH = kBlockPixels >> LOG2TW;
  constexpr int HWp = TW + 2 * HALO, HHp = TH + 2 * HALO;
  constexpr int NPIX = HWp * HHp;
  constexpr int XPIX = (TAPS == 9) ? kMaxHaloPixels : kBlockPixels;
  constexpr int X_FLOATS = XPIX * 32;
  constexpr int DY_FLOATS = kBlockPixels * 32;
  constexpr int BUF = X_FLOATS + DY_FLOATS;
  constexpr int X_ITEMS = (NPIX * 8 + kWThreads - 1) / kWThreads;
  constexpr int DY_ITEMS = (kBlockPixels * 8) / kWThreads;
  __shared__ __attribute__((aligned(16))) float buf_a[BUF];
  __shared__ __attribute__((aligned(16))) float buf_b[BUF];

  const unetpp_wgrad_desc& d = a.d;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, j = lane & 31, h = lane >> 5;  // scalar: wave-uniform LDS-DMA destinations and tests stay on the scalar unit

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

  // channels / columns that are never staged must read as zero in both buffers
  if (k_cnt < 32 || n_cnt < 32) {
    for (int i = tid; i < BUF; i += kWThreads) {
      buf_a[i] = 0.f;
      buf_b[i] = 0.f;
    }
    __syncthreads();
  }

  // ---- staging: item it covers LDS floats [it*4, it*4+4) of the image (lane-linear), pixel it>>3, quad it&7 ----
  // Per-thread byte offsets of every item relative to the tile's origin pixel are tile independent: computed once,
  // so that issuing a DMA is one scalar base (tile origin) + one VGPR offset.
  const int cc = (tid & 7) << 2;
  const bool kx_ok = cc < k_cnt, nx_ok = cc < n_cnt;
  unsigned xdelta[X_ITEMS], ydelta[DY_ITEMS];
#pragma unroll
  for (int q = 0; q < X_ITEMS; ++q) {
    const int hp = (tid >> 3) + q * (kWThreads >> 3);
    const int hy = hp / HWp, hx = hp - hy * HWp;
    xdelta[q] = static_cast<unsigned>(((hy * X.sy) * X.Ws + hx * X.sx) * X.C + cc) * 4u;
  }
#pragma unroll
  for (int q = 0; q < DY_ITEMS; ++q) {
    const int p = (tid >> 3) + q * (kWThreads >> 3);
    ydelta[q] = static_cast<unsigned>((((p >> LOG2TW) * DY.sy) * DY.Ws + (p & (TW - 1)) * DY.sx) * DY.C + cc) * 4u;
  }

  // Edge tiles take two passes: first the zero fills of out-of-image pixels (plain ds_writes), then the DMAs.  A
  // ds_write into an array with a DMA in flight makes hipcc drain vmcnt first, so no DMA may be pending when the zero
  // fills are issued -- at this point the previous tile's DMA has been waited for and this tile's has not started.
  auto issue_tile = [&](long tile, float* buf) {
    long b = tile;
    const int txi = static_cast<int>(b % a.tiles_x);
    b /= a.tiles_x;
    const int tyi = static_cast<int>(b % a.tiles_y);
    const int n = static_cast<int>(b / a.tiles_y);
    const int ty0 = tyi * TH, tx0 = txi * TW;
    // origin pixel of the staged patch (may lie outside the image; only in-image items are dereferenced)
    const char* xb = reinterpret_cast<const char*>(X.ptr + view_pixel_offset(X, n, ty0 - HALO, tx0 - HALO) + c0);
    const char* yb = reinterpret_cast<const char*>(DY.ptr + view_pixel_offset(DY, n, ty0, tx0) + nc0);
    const bool interior = ty0 >= HALO && tx0 >= HALO && ty0 + TH + HALO <= d.H && tx0 + TW + HALO <= d.W;
    if (interior) {
      if (kx_ok) {
#pragma unroll
        for (int q = 0; q < X_ITEMS; ++q) {
          float* lbase = buf + (q * kWThreads + wave * 64) * 4;  // wave-uniform; the DMA adds lane * 16 bytes
          if ((q + 1) * kWThreads <= NPIX * 8 || tid + q * kWThreads < NPIX * 8)
            __builtin_amdgcn_global_load_lds((gptr_t)(xb + xdelta[q]), (lptr_t)lbase, 16, 0, 0);
        }
      }
      if (nx_ok) {
#pragma unroll
        for (int q = 0; q < DY_ITEMS; ++q) {
          float* lbase = buf + X_FLOATS + (q * kWThreads + wave * 64) * 4;
          __builtin_amdgcn_global_load_lds((gptr_t)(yb + ydelta[q]), (lptr_t)lbase, 16, 0, 0);
        }
      }
    } else {
      unsigned xin = 0, yin = 0;  // bit q: item q of this thread lies inside the image
#pragma unroll
      for (int q = 0; q < X_ITEMS; ++q) {
        const int it = tid + q * kWThreads;
        const int hp = it >> 3;
        const int hy = hp / HWp, hx = hp - hy * HWp;
        const int y = ty0 + hy - HALO, x = tx0 + hx - HALO;
        const bool valid = it < NPIX * 8 && kx_ok;
        const bool inimg = y >= 0 && y < d.H && x >= 0 && x < d.W;
        if (valid && inimg) xin |= 1u << q;
        if (valid && !inimg) *reinterpret_cast<f32x4*>(buf + it * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int q = 0; q < DY_ITEMS; ++q) {
        const int it = tid + q * kWThreads;
        const int p = it >> 3;
        const bool inimg = ty0 + (p >> LOG2TW) < d.H && tx0 + (p & (TW - 1)) < d.W;
        if (nx_ok && inimg) yin |= 1u << q;
        if (nx_ok && !inimg) *reinterpret_cast<f32x4*>(buf + X_FLOATS + it * 4) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int q = 0; q < X_ITEMS; ++q) {
        float* lbase = buf + (q * kWThreads + wave * 64) * 4;
        if ((xin >> q) & 1u) __builtin_amdgcn_global_load_lds((gptr_t)(xb + xdelta[q]), (lptr_t)lbase, 16, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < DY_ITEMS; ++q) {
        float* lbase = buf + X_FLOATS + (q * kWThreads + wave * 64) * 4;
        if ((yin >> q) & 1u) __builtin_amdgcn_global_load_lds((gptr_t)(yb + ydelta[q]), (lptr_t)lbase, 16, 0, 0);
      }
    }
  };

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float dbsum = 0.f;

  // MFMAs of half a staged tile: wave w owns pixels 32w .. 32w+31 as 16 pairs (k-dim of the MFMA = the pixel pair)
  auto compute = [&](const float* buf, int half) {
#pragma unroll 4
    for (int pp = half * 8; pp < half * 8 + 8; ++pp) {
      const int p = 32 * wave + 2 * pp + h;
      const float bv = buf[X_FLOATS + p * 32 + j];
      dbsum += bv;
      const int xb = ((p >> LOG2TW) * HWp + (p & (TW - 1))) * 32 + j;
#pragma unroll
      for (int t = 0; t < TAPS; ++t)
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(buf[xb + ((TAPS == 9) ? ((t / 3) * HWp + (t % 3)) * 32 : 0)], bv,
                                                      acc[t], 0, 0, 0);
    }
  };

  // tiles of this workgroup: blockIdx.x, +gridDim.x, ...   (buffer A holds even, buffer B odd local tiles)
  const long stride = gridDim.x;
  const long t0 = blockIdx.x;
  const long n_my = (t0 < a.n_pix_tiles) ? (a.n_pix_tiles - t0 + stride - 1) / stride : 0;
  if (n_my > 0) issue_tile(t0, buf_a);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  // Waves w and w+4 share a SIMD.  Waves 0-3 issue the next tile's DMAs before their MFMAs, waves 4-7 between the
  // two halves of theirs, so that while one wave of a SIMD sits in address/VMEM issue the other feeds the MFMA pipe.
  const bool late = wave >= 4;
  for (long i = 0; i < n_my; i += 2) {
    if (!late && i + 1 < n_my) issue_tile(t0 + (i + 1) * stride, buf_b);  // lands while buffer A is being computed
    compute(buf_a, 0);
    if (late && i + 1 < n_my) issue_tile(t0 + (i + 1) * stride, buf_b);
    compute(buf_a, 1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (i + 1 < n_my) {
      if (!late && i + 2 < n_my) issue_tile(t0 + (i + 2) * stride, buf_a);
      compute(buf_b, 0);
      if (late && i + 2 < n_my) issue_tile(t0 + (i + 2) * stride, buf_a);
      compute(buf_b, 1);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
    }
  }

  // ---- fixed-order tree sum of the 8 waves through LDS, then one slab per workgroup ----
  tree_sum_waves<TAPS>(acc, buf_a, buf_b, wave, lane);
  const long slab_stride = (static_cast<long>(TAPS) * a.Ktot + 1) * a.Ncols;
  float* slab = d.slabs + blockIdx.x * slab_stride;
  if (wave == 0) store_slab_block<TAPS>(acc, slab, a.Ktot, a.Ncols, kbase + c0, k_cnt, n0, n_cnt, j, h);
  if (want_db) {
    dbsum += __shfl_xor(dbsum, 32);
    float* dbs = buf_b;
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

// returns UNETPP_OK after launching, or 1 when the descriptor needs another kernel (any load transform)
int launch_wgrad_dma(const unetpp_wgrad_desc* d, int Ktot, int Ncols, int n_tiles_cols, int k_tiles, hipStream_t st) {
  for (int i = 0; i < d->n_x; ++i)
    if (!plain_aligned(d->x[i])) return 1;
  for (int i = 0; i < d->n_dy; ++i)
    if (!plain_aligned(d->dy[i])) return 1;
  WDmaArgs a;
  a.d = *d;
  a.Ktot = Ktot;
  a.Ncols = Ncols;
  a.n_tiles_cols = n_tiles_cols;
  const TileGeom g = tile_geom(d->H, d->W);
  a.tiles_x = g.tiles_x;
  a.tiles_y = g.tiles_y;
  a.n_pix_tiles = static_cast<long>(d->N) * g.tiles_y * g.tiles_x;
  const dim3 grid(static_cast<unsigned>(d->n_split), static_cast<unsigned>(static_cast<long>(k_tiles) * n_tiles_cols));
  const dim3 block(kWThreads);
#define UNETPP_LAUNCH_WDMA(T)                                                                     \
  do {                                                                                            \
    if (g.log2tw == 5) hipLaunchKernelGGL((wgrad_dma_kernel<T, 5>), grid, block, 0, st, a);       \
    else if (g.log2tw == 4) hipLaunchKernelGGL((wgrad_dma_kernel<T, 4>), grid, block, 0, st, a);  \
    else hipLaunchKernelGGL((wgrad_dma_kernel<T, 3>), grid, block, 0, st, a);                     \
  } while (0)
  if (d->taps == 9) UNETPP_LAUNCH_WDMA(9);
  else UNETPP_LAUNCH_WDMA(1);
#undef UNETPP_LAUNCH_WDMA
  note_kernel(d->taps == 9 ? "wgrad_dma_kernel<9>" : "wgrad_dma_kernel<1>");
  return launch_status();
}

}  // namespace unetpp

// Fast path of the weight-gradient kernel (same math, MFMA maps and slab format as wgrad.hip) for plain,
// 16-byte aligned views.
//
//   * 512 threads = 8 waves per workgroup, one workgroup per CU (2 waves per SIMD); every wave owns 32 of the
//     tile's 256 pixels and keeps 9 taps x 16 accumulator registers live across the whole pixel loop;
//   * two LDS buffers (x patch with halo 43.5 KB + dy patch 32 KB each, 151 KB in all).  The next tile is staged
//     into the buffer the MFMAs are NOT reading, in FOUR slices interleaved with the four quarters of the MFMA
//     loop: a slice's global loads are issued before a quarter (36 MFMAs, ~2.3 k cycles) and written to LDS
//     after it, so only 12 staging registers are live instead of 40 and the MFMA loop keeps room to
//     software-pipeline its operands (the 10 LDS reads of pixel pair p+1 are issued before the 9 MFMAs of pair p).
//     One barrier per tile.  (LDS-DMA was tried: hipcc drains vmcnt before every ds_read while a DMA is in
//     flight, which serialises the tile pipeline.)
//   * the tile shape (TW = 8/16/32) is a template parameter: every LDS address of the MFMA loop is the lane's base
//     plus an immediate, no address registers.
// The 8 waves are summed through LDS in fixed order; one slab per workgroup (bitwise reproducible).
#include "common.h"
#include "wgrad_reduce.h"

namespace unetpp {
namespace {

constexpr int kWThreads = 512;

struct WFastArgs {
  unetpp_wgrad_desc d;
  int tiles_x, tiles_y;
  int Ktot, Ncols, n_tiles_cols;
  long n_pix_tiles;
};

template <int TAPS, int LOG2TW>
__global__ __launch_bounds__(kWThreads, 2) void wgrad_fast_kernel(const WFastArgs a) {
  constexpr int HALO = (TAPS == 9) ? 1 : 0;
  constexpr int TW = 1 << LOG2TW, TH = kBlockPixels >> LOG2TW;
  constexpr int HWp = TW + 2 * HALO, HHp = TH + 2 * HALO;
  constexpr int NPIX = HWp * HHp;
  constexpr int XPIX = (TAPS == 9) ? kMaxHaloPixels : kBlockPixels;
  constexpr int X_FLOATS = XPIX * 32;
  constexpr int DY_FLOATS = kBlockPixels * 32;
  constexpr int BUF = X_FLOATS + DY_FLOATS;
  constexpr int X_ITEMS = (NPIX * 8 + kWThreads - 1) / kWThreads;   // <= 6
  constexpr int DY_ITEMS = (kBlockPixels * 8) / kWThreads;          // 4
  constexpr int N_ITEMS = X_ITEMS + DY_ITEMS;
  constexpr int SLICE = (N_ITEMS + 3) / 4;                          // items per pipeline slice (<= 3)
  extern __shared__ __attribute__((aligned(16))) float smem[];      // 2 * BUF floats

  const unetpp_wgrad_desc& d = a.d;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, j = lane & 31, h = lane >> 5;  // (scalar wave index)

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

  // columns / channels that are never staged must read as zero in both buffers (full 32x32 tiles stage everything)
  if (k_cnt < 32 || n_cnt < 32) {
    for (int i = tid; i < 2 * BUF; i += kWThreads) smem[i] = 0.f;
    __syncthreads();
  }
  // optional load transform of the x view (BatchNorm apply + ReLU folded into the consumer): this workgroup's 32
  // channels' coefficients live in 64 floats of LDS behind the tile buffers
  float* coef = smem + 2 * BUF;  // [scale 32][shift 32]
  const bool x_affine = X.scale != nullptr;
  if (x_affine) {
    if (tid < 32) {
      coef[tid] = tid < k_cnt ? X.scale[c0 + tid] : 1.f;
      coef[32 + tid] = tid < k_cnt ? X.shift[c0 + tid] : 0.f;
    }
    __syncthreads();
  }

  // ---- staging items 0..X_ITEMS-1: x patch, X_ITEMS..N_ITEMS-1: dy patch.  Item geometry is recomputed from
  // tid (compile-time patch shape: the divisions are by constants). ----
  int ty0 = 0, tx0 = 0, img = 0;  // tile being staged
  auto set_tile = [&](long tile) {
    long b = tile;
    const int txi = static_cast<int>(b % a.tiles_x);
    b /= a.tiles_x;
    const int tyi = static_cast<int>(b % a.tiles_y);
    img = static_cast<int>(b / a.tiles_y);
    ty0 = tyi * TH;
    tx0 = txi * TW;
  };
  // Loads are branch-free (coordinates clamped into the image, channels past the tile clamped to 0) so that
  // hipcc keeps them in flight across the MFMA quarter; the zeroing happens at the LDS write.
  auto in_place = [](int v) {  // a thread index "produced" where it is used: item coordinates are not hoisted out of the tile loop
    asm volatile("" : "+v"(v));
    return v;
  };
  auto load_item = [&](int q) -> f32x4 {
    const int tid = in_place(static_cast<int>(threadIdx.x));
    if (q < X_ITEMS) {
      const int it = tid + q * kWThreads;
      const int hp = min(it >> 3, NPIX - 1), cc = (it & 7) << 2;
      const int hy = hp / HWp, hx = hp - hy * HWp;
      const int y = min(max(ty0 + hy - HALO, 0), d.H - 1), x = min(max(tx0 + hx - HALO, 0), d.W - 1);
      return *reinterpret_cast<const f32x4*>(X.ptr + view_pixel_offset(X, img, y, x) + c0 + (cc < k_cnt ? cc : 0));
    }
    const int it = tid + (q - X_ITEMS) * kWThreads;
    const int p = it >> 3, cc = (it & 7) << 2;
    const int y = min(ty0 + (p >> LOG2TW), d.H - 1), x = min(tx0 + (p & (TW - 1)), d.W - 1);
    return *reinterpret_cast<const f32x4*>(DY.ptr + view_pixel_offset(DY, img, y, x) + nc0 + (cc < n_cnt ? cc : 0));
  };
  auto store_item = [&](int q, float* buf, f32x4 v) {
    if (q < X_ITEMS) {
      const int it = tid + q * kWThreads;
      const int hp = it >> 3;
      const int hy = hp / HWp, hx = hp - hy * HWp;
      const int y = ty0 + hy - HALO, x = tx0 + hx - HALO;
      const bool keep = y >= 0 && y < d.H && x >= 0 && x < d.W;
      if (x_affine) {
        const f32x4 sc = *reinterpret_cast<const f32x4*>(&coef[(it & 7) << 2]);
        const f32x4 sh = *reinterpret_cast<const f32x4*>(&coef[32 + ((it & 7) << 2)]);
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
      }
      if (X.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = keep ? v[e] : 0.f;
      if (it < NPIX * 8 && ((it & 7) << 2) < k_cnt) *reinterpret_cast<f32x4*>(&buf[it * 4]) = v;
    } else {
      const int it = tid + (q - X_ITEMS) * kWThreads;
      const int p = it >> 3;
      const bool keep = ty0 + (p >> LOG2TW) < d.H && tx0 + (p & (TW - 1)) < d.W;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = keep ? v[e] : 0.f;
      if (((it & 7) << 2) < n_cnt) *reinterpret_cast<f32x4*>(&buf[X_FLOATS + it * 4]) = v;
    }
  };

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float dbsum = 0.f;

  // operands of one pixel pair: B = dy[p][col j], A[t] = x[p (+) tap t][channel j]; p = 32*wave + 2*pp + h
  struct Ops {
    float b;
    float a[TAPS];
  };
  const int p_base = 32 * wave + h;
  auto read_ops = [&](const float* buf, int pp) {
    const int p = p_base + 2 * pp;
    Ops o;
    o.b = buf[X_FLOATS + p * 32 + j];
    const int xb = ((p >> LOG2TW) * HWp + (p & (TW - 1))) * 32 + j;
#pragma unroll
    for (int t = 0; t < TAPS; ++t) o.a[t] = buf[xb + ((TAPS == 9) ? ((t / 3) * HWp + (t % 3)) * 32 : 0)];
    return o;
  };

  // tiles of this workgroup: blockIdx.x, +gridDim.x, ...
  const long stride = gridDim.x;
  const long t0 = blockIdx.x;
  const long n_my = (t0 < a.n_pix_tiles) ? (a.n_pix_tiles - t0 + stride - 1) / stride : 0;
  if (n_my > 0) {
    set_tile(t0);
#pragma unroll
    for (int q = 0; q < N_ITEMS; ++q) store_item(q, smem, load_item(q));
  }
  __syncthreads();
  for (long i = 0; i < n_my; ++i) {
    const float* cur = smem + (i & 1) * BUF;
    float* nxt = smem + ((i + 1) & 1) * BUF;  // last read in iteration i-1 (barrier at its end)
    const bool more = i + 1 < n_my;
    if (more) set_tile(t0 + (i + 1) * stride);
    Ops o = read_ops(cur, 0);
#pragma unroll
    for (int quarter = 0; quarter < 4; ++quarter) {
      f32x4 stage[SLICE];
      if (more) {
#pragma unroll
        for (int u = 0; u < SLICE; ++u)
          if (quarter * SLICE + u < N_ITEMS) stage[u] = load_item(quarter * SLICE + u);
      }
#pragma unroll
      for (int pq = 0; pq < 4; ++pq) {
        const int pp = quarter * 4 + pq;
        Ops nx = o;
        if (pp + 1 < 16) nx = read_ops(cur, pp + 1);
        // (pinning "next pair's reads before this pair's MFMAs" with sched_barrier measured 2 % slower than
        // hipcc's own order: with two waves per SIMD the LDS latency is already covered)
        dbsum += o.b;
#pragma unroll
        for (int t = 0; t < TAPS; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(o.a[t], o.b, acc[t], 0, 0, 0);
        o = nx;
      }
      if (more) {
#pragma unroll
        for (int u = 0; u < SLICE; ++u)
          if (quarter * SLICE + u < N_ITEMS) store_item(quarter * SLICE + u, nxt, stage[u]);
      }
    }
    __syncthreads();
  }

  // ---- fixed-order tree sum of the 8 waves through LDS, then one slab per workgroup ----
  tree_sum_waves<TAPS>(acc, smem, smem + 2 * TAPS * 1024, wave, lane);
  const long slab_stride = (static_cast<long>(TAPS) * a.Ktot + 1) * a.Ncols;
  float* slab = d.slabs + blockIdx.x * slab_stride;
  if (wave == 0) store_slab_block<TAPS>(acc, slab, a.Ktot, a.Ncols, kbase + c0, k_cnt, n0, n_cnt, j, h);
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

// x views may carry an affine + ReLU load transform (folded BatchNorm); ReLU gates need the generic kernel
bool aligned_view(const unetpp_view& v, bool allow_affine) {
  if (v.gate != nullptr) return false;
  if (!allow_affine && (v.scale != nullptr || v.relu)) return false;
  return ((v.C | v.c_off | v.c_len) & 3) == 0 && (reinterpret_cast<uintptr_t>(v.ptr) & 15) == 0;
}

template <int TAPS, int LOG2TW>
int launch_one(const WFastArgs& a, dim3 grid, hipStream_t st) {
  constexpr size_t lds =
      (2 * (((TAPS == 9) ? kMaxHaloPixels : kBlockPixels) * 32 + kBlockPixels * 32) + 64) * sizeof(float);
  // > 64 KB of dynamic LDS needs the per-function opt-in; it is idempotent and keeps the ABI stateless
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_fast_kernel<TAPS, LOG2TW>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess)
    return UNETPP_ELAUNCH;
  hipLaunchKernelGGL((wgrad_fast_kernel<TAPS, LOG2TW>), grid, dim3(kWThreads), lds, st, a);
  note_kernel(TAPS == 9 ? "wgrad_fast_kernel<9>" : "wgrad_fast_kernel<1>");
  return launch_status();
}

}  // namespace

// returns UNETPP_OK after launching, or 1 when the descriptor needs the generic kernel
int launch_wgrad_fast(const unetpp_wgrad_desc* d, int Ktot, int Ncols, int n_tiles_cols, int k_tiles, hipStream_t st) {
  for (int i = 0; i < d->n_x; ++i)
    if (!aligned_view(d->x[i], true)) return 1;
  for (int i = 0; i < d->n_dy; ++i)
    if (!aligned_view(d->dy[i], false)) return 1;
  WFastArgs a;
  a.d = *d;
  a.Ktot = Ktot;
  a.Ncols = Ncols;
  a.n_tiles_cols = n_tiles_cols;
  const TileGeom g = tile_geom(d->H, d->W);
  a.tiles_x = g.tiles_x;
  a.tiles_y = g.tiles_y;
  a.n_pix_tiles = static_cast<long>(d->N) * g.tiles_y * g.tiles_x;
  const dim3 grid(static_cast<unsigned>(d->n_split), static_cast<unsigned>(static_cast<long>(k_tiles) * n_tiles_cols));
  if (d->taps == 9) {
    if (g.log2tw == 5) return launch_one<9, 5>(a, grid, st);
    if (g.log2tw == 4) return launch_one<9, 4>(a, grid, st);
    return launch_one<9, 3>(a, grid, st);
  }
  if (g.log2tw == 5) return launch_one<1, 5>(a, grid, st);
  if (g.log2tw == 4) return launch_one<1, 4>(a, grid, st);
  return launch_one<1, 3>(a, grid, st);
}

}  // namespace unetpp

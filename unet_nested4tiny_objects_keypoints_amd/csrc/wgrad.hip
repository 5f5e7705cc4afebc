// Weight gradient of the multi-view pixel GEMM on the gfx950 matrix cores.
//
//   dW[tap][k][n] = sum_p x[p (+) tap, k] * dy[p, n]        db[n] = sum_p dy[p, n]
//
// The reduction runs over ALL pixels of the batch (2 M at 256x256x32), the output is tiny, so the
// kernel is split-K over pixel tiles: blockIdx.y picks a (32 k-channels, 32 columns) output tile,
// blockIdx.x is one of n_split workers that strides over the 256-pixel tiles.  Per tile the x patch
// (with halo, load transform applied) and the dy patch (ReLU gate applied) go to LDS; each wave owns
// 64 pixels and issues one v_mfma_f32_32x32x2_f32 per (pixel pair, tap) with A = x (row = k channel,
// k-dim = pixel) and B = dy (k-dim = pixel, col = n).  9 taps x 16 accumulator registers stay live for
// the whole pixel loop; the 4 waves are then summed through LDS in fixed order and the block writes
// one partial slab, so the result is bitwise reproducible (no atomics).
#include "common.h"

namespace unetpp {
namespace {

struct WgradArgs {
  unetpp_wgrad_desc d;
  int log2tw, tiles_x, tiles_y;
  int Ktot, Ncols, n_tiles_cols;
  long n_pix_tiles;
};

template <int TAPS>
__global__ __launch_bounds__(kThreads, 2) void wgrad_kernel(const WgradArgs a) {
  constexpr int HALO = (TAPS == 9) ? 1 : 0;
  constexpr int X_FLOATS = (TAPS == 9 ? kMaxHaloPixels : kBlockPixels) * 32;
  constexpr int DY_FLOATS = kBlockPixels * 32;
  constexpr int RED_FLOATS = TAPS * 32 * 32;
  constexpr int SM0 = X_FLOATS > RED_FLOATS ? X_FLOATS : RED_FLOATS;
  __shared__ __attribute__((aligned(16))) float smem[SM0 + DY_FLOATS];
  float* x_tile = smem;
  float* dy_tile = smem + SM0;

  const unetpp_wgrad_desc& d = a.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;

  // ---- output tile: 32 k-channels of one x view, 32 columns of dy ----
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
  const int nc0 = nt * 32;             // first channel inside the dy view
  const int n0 = col_base + nc0;       // first GEMM column
  const int n_cnt = min(32, DY.c_len - nc0);
  const bool xvec = view_vec4(X);
  const bool dvec = view_vec4(DY);
  const bool want_db = (blockIdx.y / a.n_tiles_cols) == 0;

  const int TW = 1 << a.log2tw, TH = kBlockPixels >> a.log2tw;
  const int HWp = TW + 2 * HALO, HHp = TH + 2 * HALO;
  const int npix = HWp * HHp;

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float dbsum = 0.f;

  for (long tile = blockIdx.x; tile < a.n_pix_tiles; tile += gridDim.x) {
    long b = tile;
    const int txi = static_cast<int>(b % a.tiles_x);
    b /= a.tiles_x;
    const int tyi = static_cast<int>(b % a.tiles_y);
    const int n = static_cast<int>(b / a.tiles_y);
    const int ty0 = tyi * TH, tx0 = txi * TW;

    __syncthreads();
    for (int it = tid; it < npix * 8; it += kThreads) {
      const int hp = it >> 3, cc = (it & 7) * 4;
      const int hy = hp / HWp, hx = hp - hy * HWp;
      const int y = ty0 + hy - HALO, x = tx0 + hx - HALO;
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if (cc < k_cnt && y >= 0 && y < d.H && x >= 0 && x < d.W)
        val = view_load4(X, view_pixel_offset(X, n, y, x), c0 + cc, k_cnt - cc, xvec);
      *reinterpret_cast<f32x4*>(&x_tile[hp * 32 + cc]) = val;
    }
    for (int it = tid; it < kBlockPixels * 8; it += kThreads) {
      const int p = it >> 3, cc = (it & 7) * 4;
      const int y = ty0 + (p >> a.log2tw), x = tx0 + (p & (TW - 1));
      f32x4 val = {0.f, 0.f, 0.f, 0.f};
      if (cc < n_cnt && y < d.H && x < d.W)
        val = view_load4(DY, view_pixel_offset(DY, n, y, x), nc0 + cc, n_cnt - cc, dvec);
      *reinterpret_cast<f32x4*>(&dy_tile[p * 32 + cc]) = val;
    }
    __syncthreads();

#pragma unroll 4
    for (int pp = 0; pp < 32; ++pp) {
      const int p = 64 * wave + 2 * pp + h;
      const float bv = dy_tile[p * 32 + j];
      dbsum += bv;
      const int xb = ((p >> a.log2tw) * HWp + (p & (TW - 1))) * 32 + j;
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const int toff = (TAPS == 9) ? ((t / 3) * HWp + (t % 3)) * 32 : 0;
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(x_tile[xb + toff], bv, acc[t], 0, 0, 0);
      }
    }
  }

  // ---- fixed-order sum of the 4 waves through LDS, then one slab per block ----
  float* red = smem;  // [TAPS][32 k][32 n]
  for (int w = 0; w < 4; ++w) {
    __syncthreads();
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
  }
  __syncthreads();
  const long slab_stride = (static_cast<long>(TAPS) * a.Ktot + 1) * a.Ncols;
  float* slab = d.slabs + blockIdx.x * slab_stride;
  for (int it = tid; it < TAPS * 32 * 32; it += kThreads) {
    const int col = it & 31, row = (it >> 5) & 31, t = it >> 10;
    if (row < k_cnt && col < n_cnt)
      slab[(static_cast<long>(t) * a.Ktot + kbase + c0 + row) * a.Ncols + n0 + col] = red[it];
  }
  if (want_db) {
    dbsum += __shfl_xor(dbsum, 32);
    __syncthreads();
    if (h == 0) dy_tile[wave * 32 + j] = dbsum;
    __syncthreads();
    if (tid < n_cnt)
      slab[static_cast<long>(TAPS) * a.Ktot * a.Ncols + n0 + tid] =
          dy_tile[tid] + dy_tile[32 + tid] + dy_tile[64 + tid] + dy_tile[96 + tid];
  }
}

__global__ void wgrad_finish_kernel(const float* __restrict__ slabs, int n_split, int taps, int K, int Ncols,
                                    int n_inner, float* __restrict__ dw, long d_t, long d_k, long d_n, long d_o,
                                    float* __restrict__ db, int log2_sg) {
  // 1024 threads = EPB consecutive slab elements x SG slab groups (SG = 2^log2_sg <= 16, EPB = 1024 / SG): group g sums
  // slabs g, g + SG, ...; the groups are combined through LDS in fixed order -> reproducible.  Few slabs (wide layers:
  // many (channel tile, column tile) pairs, n_split = 2..8) take few groups and more elements per block instead of
  // leaving 7/8 of the threads idle; the value of every sum is that of the 16-group form (empty groups add zeros).
  const int SG = 1 << log2_sg, EPB = 1024 >> log2_sg;
  __shared__ float part[1024];  // [SG][EPB]
  const long rows = static_cast<long>(taps) * K + 1;
  const long total = rows * Ncols;
  const int e = threadIdx.x & (EPB - 1), g = threadIdx.x >> (10 - log2_sg);
  const long base_blocks = (total + EPB - 1) / EPB;
  if (static_cast<long>(blockIdx.x) >= base_blocks) {
    // extra blocks, 2x2 deconvolution only (n_inner < Ncols): the bias gradient of an output channel is the sum of
    // its (up to 4) pixel-phase columns -- 16 channels x 4 phases per block (64 threads per slab group), fixed order
    const int n_outer = Ncols / n_inner;
    const int o = e >> 4;
    const long nn = (blockIdx.x - base_blocks) * 16L + (e & 15);
    float u = 0.f;
    if (e < 64 && o < n_outer && nn < n_inner)
      for (int b = g; b < n_split; b += SG) u += slabs[static_cast<long>(b) * total + (rows - 1) * Ncols + o * n_inner + nn];
    part[g * EPB + e] = u;
    __syncthreads();
    if (g == 0 && e < 16 && nn < n_inner && db != nullptr) {
      float t = 0.f;
      for (int oo = 0; oo < n_outer; ++oo) {
        float v = 0.f;
        for (int q = 0; q < SG; ++q) v += part[q * EPB + oo * 16 + e];
        t += v;
      }
      db[nn] = t;
    }
    return;
  }
  const long i = blockIdx.x * static_cast<long>(EPB) + e;
  float s = 0.f;
  if (i < total) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = g;
    for (; b + 3 * SG < n_split; b += 4 * SG) {
      s0 += slabs[static_cast<long>(b) * total + i];
      s1 += slabs[static_cast<long>(b + SG) * total + i];
      s2 += slabs[static_cast<long>(b + 2 * SG) * total + i];
      s3 += slabs[static_cast<long>(b + 3 * SG) * total + i];
    }
    for (; b < n_split; b += SG) s0 += slabs[static_cast<long>(b) * total + i];
    s = (s0 + s1) + (s2 + s3);
  }
  part[g * EPB + e] = s;
  __syncthreads();
  if (g != 0 || i >= total) return;
  s = 0.f;
  for (int q = 0; q < SG; ++q) s += part[q * EPB + e];  // fixed order
  const long row = i / Ncols;
  const int nn = static_cast<int>(i - row * Ncols);
  if (row == rows - 1) {  // bias gradient of a plain convolution: the sum just formed (deconvolution: extra blocks)
    if (db != nullptr && n_inner == Ncols) db[nn] = s;
    return;
  }
  if (dw == nullptr) return;
  const long t = row / K, k = row - t * K;
  dw[t * d_t + k * d_k + (nn % n_inner) * d_n + (nn / n_inner) * d_o] = s;
}

// Wide plain convolutions (K * Ncols >= 64 K elements, dense torch layout dw[n][k][t]): in the kernel above
// neighbouring threads hold neighbouring COLUMNS n, whose destinations are 4 * taps * K bytes apart -- every 4-byte
// store its own cache line (a 256 -> 256 layer: 37 us for 2.4 MB of output).  Here a block owns 8 channels x 32
// columns x all taps, sums the slabs with 128-byte row reads (SG <= 4 slab groups of 256 threads, combined in fixed
// order: reproducible) and transposes through LDS, so that every column's 8 * taps values leave as one contiguous run.
constexpr int kTrK = 8, kTrN = 32;
__global__ __launch_bounds__(1024) void wgrad_finish_tr_kernel(const float* __restrict__ slabs, int n_split, int taps,
                                                               int K, int Ncols, float* __restrict__ dw,
                                                               float* __restrict__ db, int log2_sg) {
  constexpr int EMAX = 9 * kTrK * kTrN;
  __shared__ float part[4][EMAX];
  __shared__ float out_t[kTrN][9 * kTrK + 1];
  __shared__ float part_db[4][kTrN];
  const int SG = 1 << log2_sg;
  const int g = threadIdx.x >> 8, e0 = threadIdx.x & 255;
  const int n_blocks = Ncols / kTrN;
  const int n0 = (blockIdx.x % n_blocks) * kTrN, k0 = (blockIdx.x / n_blocks) * kTrK;
  const long total = (static_cast<long>(taps) * K + 1) * Ncols;
  const int E = taps * kTrK * kTrN;
  auto sum_slabs = [&](long addr) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = g;
    for (; b + 3 * SG < n_split; b += 4 * SG) {
      s0 += slabs[static_cast<long>(b) * total + addr];
      s1 += slabs[static_cast<long>(b + SG) * total + addr];
      s2 += slabs[static_cast<long>(b + 2 * SG) * total + addr];
      s3 += slabs[static_cast<long>(b + 3 * SG) * total + addr];
    }
    for (; b < n_split; b += SG) s0 += slabs[static_cast<long>(b) * total + addr];
    return (s0 + s1) + (s2 + s3);
  };
  for (int idx = e0; idx < E; idx += 256) {
    const int n = idx & (kTrN - 1), row = idx >> 5;  // row = t * kTrK + kk
    const int t = row / kTrK, kk = row - t * kTrK;
    part[g][idx] = sum_slabs((static_cast<long>(t) * K + k0 + kk) * Ncols + n0 + n);
  }
  const bool with_db = k0 == 0 && db != nullptr;
  if (with_db && e0 < kTrN) part_db[g][e0] = sum_slabs(static_cast<long>(taps) * K * Ncols + n0 + e0);
  __syncthreads();
  if (g == 0) {
    for (int idx = e0; idx < E; idx += 256) {
      float s = 0.f;
      for (int q = 0; q < SG; ++q) s += part[q][idx];  // fixed order
      const int n = idx & (kTrN - 1), row = idx >> 5;
      const int t = row / kTrK, kk = row - t * kTrK;
      out_t[n][kk * taps + t] = s;
    }
    if (with_db && e0 < kTrN) {
      float s = 0.f;
      for (int q = 0; q < SG; ++q) s += part_db[q][e0];
      db[n0 + e0] = s;
    }
  }
  __syncthreads();
  const int run = kTrK * taps;  // contiguous floats per column
  for (int idx = threadIdx.x; idx < E; idx += blockDim.x) {
    const int n = idx / run, r = idx - n * run;
    dw[(static_cast<long>(n0 + n) * K + k0) * taps + r] = out_t[n][r];
  }
}

__global__ void pack_weight_kernel(float* __restrict__ dst, const float* __restrict__ src, int T, int K, int Ncols,
                                   long d_t, long d_k, long d_n, long s_t, long s_k, long s_n, int flip) {
  const long total = static_cast<long>(T) * K * Ncols;
  const long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
  if (i >= total) return;
  const int nn = static_cast<int>(i % Ncols);
  const long r = i / Ncols;
  const int k = static_cast<int>(r % K);
  const int t = static_cast<int>(r / K);
  const int tt = flip ? (T - 1 - t) : t;
  dst[t * d_t + k * d_k + nn * d_n] = src[tt * s_t + k * s_k + nn * s_n];
}

// Many small strided copies in ONE launch (grid.y = job): dst[o * dst_stride + i] = src[o * src_stride + i], o < n_outer,
// i < n_inner.  The host-side re-layouts a pass needs before its GEMMs -- the four-phase bias rows of the transposed
// convolutions (src_stride 0), the grouped input-gradient weights (slices of the consumers' conv1 weights, concatenated
// along the output-channel axis) -- were one 4-5 us launch each (7 + 6 per step at depth 4, 10 + 10 at depth 5).
__global__ __launch_bounds__(256) void copy_jobs_kernel(const unetpp_copy_job* __restrict__ jobs) {
  const unetpp_copy_job j = jobs[blockIdx.y];
  const unsigned inner = static_cast<unsigned>(j.n_inner);
  const bool vec = ((j.n_inner | j.src_stride | j.dst_stride) & 3) == 0 &&
                   ((reinterpret_cast<uintptr_t>(j.src) | reinterpret_cast<uintptr_t>(j.dst)) & 15) == 0;
  const unsigned step = gridDim.x * 256u;
  if (vec) {
    const unsigned inner4 = inner >> 2, total4 = static_cast<unsigned>(j.n_outer) * inner4;
    for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total4; e += step) {
      const unsigned o = e / inner4, i = e - o * inner4;
      *reinterpret_cast<f32x4*>(j.dst + o * j.dst_stride + 4 * i) = *reinterpret_cast<const f32x4*>(j.src + o * j.src_stride + 4 * i);
    }
  } else {
    const unsigned total = static_cast<unsigned>(j.n_outer) * inner;
    for (unsigned e = blockIdx.x * 256u + threadIdx.x; e < total; e += step) {
      const unsigned o = e / inner, i = e - o * inner;
      j.dst[o * j.dst_stride + i] = j.src[o * j.src_stride + i];
    }
  }
}

}  // namespace
}  // namespace unetpp

using namespace unetpp;

extern "C" int32_t unetpp_wgrad_max_split(int32_t N, int32_t H, int32_t W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  const TileGeom g = tile_geom(H, W);
  const int64_t t = static_cast<int64_t>(N) * g.tiles_y * g.tiles_x;
  return static_cast<int32_t>(t > 4096 ? 4096 : t);
}

extern "C" int32_t unetpp_wgrad_slab_planes(const unetpp_wgrad_desc* d) {
  if (d == nullptr) return 0;
  // the 1..4-channel first layer keeps its own kernel (tap slabs) whatever the flags say
  const bool small = d->taps == 9 && d->n_x == 1 && d->x[0].c_len <= 4;
  if (d->flags & UNETPP_GEMM_BF16) return d->taps;  // bf16 storage: direct summation only
  return (!small && wgrad_wino_applies(d)) ? 16 : d->taps;
}

extern "C" int32_t unetpp_wgrad_pairs_per_workgroup(const unetpp_wgrad_desc* d) {
  if (d == nullptr) return 0;
  const int pw = wgrad_pw_pairs(d);
  if (pw > 0) return pw;
  return wgrad_bf16_quads(d) ? 4 : 1;
}

extern "C" int unetpp_wgrad(const unetpp_wgrad_desc* d, void* stream) {
  if (d == nullptr || d->N <= 0 || d->H <= 0 || d->W <= 0) return UNETPP_EINVAL;
  if (d->taps != 9 && d->taps != 1) return UNETPP_EINVAL;
  if (d->n_x < 1 || d->n_x > UNETPP_MAX_VIEWS || d->slabs == nullptr) return UNETPP_EINVAL;
  if (d->n_dy < 1 || d->n_dy > UNETPP_MAX_VIEWS) return UNETPP_EINVAL;
  WgradArgs a;
  a.d = *d;
  a.Ktot = 0;
  int k_tiles = 0;
  for (int i = 0; i < d->n_x; ++i) {
    if (!view_ok(d->x[i]) || !view_covers(d->x[i], d->H, d->W)) return UNETPP_EINVAL;
    a.Ktot += d->x[i].c_len;
    k_tiles += (d->x[i].c_len + 31) / 32;
  }
  a.Ncols = 0;
  a.n_tiles_cols = 0;
  for (int i = 0; i < d->n_dy; ++i) {
    if (!view_ok(d->dy[i]) || !view_covers(d->dy[i], d->H, d->W)) return UNETPP_EINVAL;
    a.Ncols += d->dy[i].c_len;
    a.n_tiles_cols += (d->dy[i].c_len + 31) / 32;
  }
  const TileGeom g = tile_geom(d->H, d->W);
  a.log2tw = g.log2tw;
  a.tiles_x = g.tiles_x;
  a.tiles_y = g.tiles_y;
  a.n_pix_tiles = static_cast<long>(d->N) * g.tiles_y * g.tiles_x;
  if (d->n_split < 1 || d->n_split > a.n_pix_tiles || d->n_split > 4096) return UNETPP_EINVAL;
  const long pairs = static_cast<long>(k_tiles) * a.n_tiles_cols;
  if (pairs > 65535) return UNETPP_EINVAL;
  const dim3 grid(static_cast<unsigned>(d->n_split), static_cast<unsigned>(pairs));
  hipStream_t st = static_cast<hipStream_t>(stream);
  const int small = launch_small_cin_wgrad(d, st);  // 1..4-channel first layer
  if (small != 1) return small;
  if (d->flags & UNETPP_GEMM_BF16) return launch_wgrad_bf16(d, a.Ktot, a.Ncols, a.n_tiles_cols, k_tiles, st);
  {
    const int pw = launch_wgrad_pw(d, st);  // plain pointwise launches in 64 x 128 blocks: operands straight into registers
    if (pw != 1) return pw;
  }
  {
    const int wino = launch_wgrad_wino(d, a.Ktot, a.Ncols, a.n_tiles_cols, k_tiles, st);  // 16-plane slabs
    if (wino != 1) return wino;
  }
  {
    const int dma = launch_wgrad_dma(d, a.Ktot, a.Ncols, a.n_tiles_cols, k_tiles, st);  // plain views, direct sum
    if (dma != 1) return dma;
  }
  const int fast = launch_wgrad_fast(d, a.Ktot, a.Ncols, a.n_tiles_cols, k_tiles, st);
  if (fast != 1) return fast;  // launched (or failed to); 1 = views need the generic kernel
  if (d->taps == 9)
    hipLaunchKernelGGL(wgrad_kernel<9>, grid, dim3(kThreads), 0, st, a);
  else
    hipLaunchKernelGGL(wgrad_kernel<1>, grid, dim3(kThreads), 0, st, a);
  note_kernel(d->taps == 9 ? "wgrad_kernel<9>" : "wgrad_kernel<1>");
  return launch_status();
}

extern "C" int unetpp_wgrad_finish(const float* slabs, int32_t n_split, int32_t taps, int32_t K, int32_t Ncols,
                                   int32_t n_inner, float* dw, int64_t d_t, int64_t d_k, int64_t d_n, int64_t d_o,
                                   float* db, void* stream) {
  if (slabs == nullptr || n_split < 1 || taps < 1 || K < 1 || Ncols < 1) return UNETPP_EINVAL;
  if (n_inner < 1 || Ncols % n_inner != 0) return UNETPP_EINVAL;
  if (taps == 16) {  // Winograd-domain slabs (unetpp_wgrad_slab_planes() == 16): sum, then G^T . G -> 9 taps
    if (n_inner != Ncols) return UNETPP_EINVAL;
    return launch_wgrad_finish_wino(slabs, n_split, K, Ncols, dw, d_t, d_k, d_n, db, static_cast<hipStream_t>(stream));
  }
  const long total = (static_cast<long>(taps) * K + 1) * Ncols;
  if (Ncols / n_inner > 4) return UNETPP_EINVAL;  // at most 4 pixel phases per output channel
  if (n_inner == Ncols && dw != nullptr && taps <= 9 && d_t == 1 && d_k == taps && d_n == static_cast<int64_t>(K) * taps &&
      K % kTrK == 0 && Ncols % kTrN == 0 && static_cast<long>(K) * Ncols >= 65536 && n_split <= 32) {
    int log2_tr = 0;
    while ((1 << log2_tr) < n_split && log2_tr < 2) ++log2_tr;
    const unsigned tr_blocks = static_cast<unsigned>((K / kTrK) * (Ncols / kTrN));
    hipLaunchKernelGGL(wgrad_finish_tr_kernel, dim3(tr_blocks), dim3(256u << log2_tr), 0, static_cast<hipStream_t>(stream),
                       slabs, n_split, taps, K, Ncols, dw, db, log2_tr);
    return launch_status();
  }
  const int log2_sg = finish_log2_groups(n_split);
  const long epb = 1024 >> log2_sg;
  const unsigned blocks = static_cast<unsigned>((total + epb - 1) / epb + (n_inner != Ncols ? (n_inner + 15) / 16 : 0));
  hipLaunchKernelGGL(wgrad_finish_kernel, dim3(blocks), dim3(1024), 0, static_cast<hipStream_t>(stream), slabs,
                     n_split, taps, K, Ncols, n_inner, dw, d_t, d_k, d_n, d_o, db, log2_sg);
  return launch_status();
}

extern "C" int unetpp_pack_weight(float* dst, const float* src, int32_t T, int32_t K, int32_t Ncols, int64_t d_t,
                                  int64_t d_k, int64_t d_n, int64_t s_t, int64_t s_k, int64_t s_n, int32_t flip,
                                  void* stream) {
  if (dst == nullptr || src == nullptr || T < 1 || K < 1 || Ncols < 1) return UNETPP_EINVAL;
  const long total = static_cast<long>(T) * K * Ncols;
  const unsigned blocks = static_cast<unsigned>((total + 255) / 256);
  hipLaunchKernelGGL(pack_weight_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), dst, src, T,
                     K, Ncols, d_t, d_k, d_n, s_t, s_k, s_n, flip);
  return launch_status();
}

extern "C" int unetpp_copy_jobs(const unetpp_copy_job* jobs_device, int32_t n_jobs, int64_t max_elems, void* stream) {
  // (the jobs live in device memory: their fields are the caller's contract -- n_outer * n_inner < 2^31 per job)
  if (jobs_device == nullptr || n_jobs < 1 || n_jobs > 65535 || max_elems < 1 || max_elems >= 0x7fffffffLL) return UNETPP_EINVAL;
  const int64_t want = (max_elems + 1023) / 1024;  // 256 threads x one 16-byte piece
  hipLaunchKernelGGL(copy_jobs_kernel, dim3(static_cast<unsigned>(want < 512 ? want : 512), static_cast<unsigned>(n_jobs)),
                     dim3(256), 0, static_cast<hipStream_t>(stream), jobs_device);
  return launch_status();
}

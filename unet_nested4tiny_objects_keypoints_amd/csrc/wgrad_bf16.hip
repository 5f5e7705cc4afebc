// bf16-storage twin of the weight-gradient kernel (UNETPP_GEMM_BF16 in unetpp_wgrad_desc.flags): x and dy are bf16
// NHWC in HBM, the products run on v_mfma_f32_32x32x16_bf16, the sums (accumulators, slabs, dW, db) are fp32.
//
//   dW[tap][k][n] = sum_p x[p (+) tap, k] * dy[p, n]       MFMA row = input channel k, column = output column n,
//                                                           MFMA k = 16 PIXELS
// Both operands are needed with the pixel index along the MFMA k dimension, i.e. transposed against the NHWC tiles
// ([pixel][32 channels], 64-byte rows) that the staging writes into LDS.  gfx950's ds_read_b64_tr_b16 does that
// transpose in the read: a 16-lane group supplies the addresses of a block of 4 pixel rows x 16 channels (lane 4q+p:
// row q, channels 4p..4p+3) and lane i receives channel i of the 4 rows.  Two such reads give a lane the 8 pixels
// 8h..8h+7 of its channel -- exactly the A (rows = channels) or B (columns = channels) fragment of the 32x32x16 MFMA.
// The 4 pixel rows of a block are 4 horizontally adjacent pixels (256 contiguous bytes: all 64 banks, conflict free),
// and a tap only shifts the block: every x address is the lane's base plus an immediate.
//
// Skeleton as wgrad_fast.hip: 512 threads = 8 waves, one workgroup per CU; a wave owns 32 of the tile's 256 pixels
// (two 16-pixel MFMA steps) and keeps 9 taps x 16 accumulator registers over its whole pixel loop; two LDS buffers,
// the next tile's global loads are issued before the MFMAs of the current tile and written to LDS after them (the
// BatchNorm-apply + ReLU load transform of x runs there, in fp32, rounded to bf16); fixed-order tree sum of the 8
// waves; one slab per workgroup in the format of wgrad.hip ([taps*K + 1][Ncols] fp32, last row = db).
#include "bf16_common.h"
#include "common.h"
#include "wgrad_reduce.h"

namespace unetpp {
namespace {

constexpr int kWThreads = 512;

struct WBfArgs {
  unetpp_wgrad_desc d;
  int tiles_x, tiles_y;
  int Ktot, Ncols, n_tiles_cols;
  long n_pix_tiles;
};

template <int OFF>
__device__ __forceinline__ void lds_read_tr(u32x2& v, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_wait2(u32x2& a, u32x2& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); }

template <int TAPS, int LOG2TW>
__global__ __launch_bounds__(kWThreads, 1) void wgrad_bf16_kernel(const WBfArgs a) {
  constexpr int HALO = (TAPS == 9) ? 1 : 0;
  constexpr int TW = 1 << LOG2TW, TH = kBlockPixels >> LOG2TW;
  constexpr int HWp = TW + 2 * HALO, HHp = TH + 2 * HALO;
  constexpr int NPIX = HWp * HHp;
  constexpr int XPIX = (TAPS == 9) ? kMaxHaloPixels : kBlockPixels;
  constexpr int X_BYTES = XPIX * 64;
  constexpr int DY_BYTES = kBlockPixels * 64;
  constexpr int BUF = X_BYTES + DY_BYTES;
  constexpr int X_ITEMS = (NPIX * 4 + kWThreads - 1) / kWThreads;  // 16-byte items (8 channels), 4 per pixel: <= 3
  constexpr int DY_ITEMS = (kBlockPixels * 4) / kWThreads;         // 2
  constexpr int N_ITEMS = X_ITEMS + DY_ITEMS;
  constexpr int TREE_BYTES = 4 * TAPS * 4096;                      // four regions of TAPS*1024 floats
  constexpr int TILE_BYTES = (2 * BUF > TREE_BYTES) ? 2 * BUF : TREE_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // TILE_BYTES + 256

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
  const bf16_t* xptr = reinterpret_cast<const bf16_t*>(X.ptr);
  const bf16_t* dyptr = reinterpret_cast<const bf16_t*>(DY.ptr);
  const int c0 = kt * 32;
  const int k_cnt = min(32, X.c_len - c0);
  const int nc0 = nt * 32;
  const int n0 = col_base + nc0;
  const int n_cnt = min(32, DY.c_len - nc0);
  const bool want_db = (blockIdx.y / a.n_tiles_cols) == 0;

  // channels / columns that are never staged must read as zero in both buffers
  if (k_cnt < 32 || n_cnt < 32) {
    for (int i = tid; i < 2 * BUF / 4; i += kWThreads) reinterpret_cast<unsigned*>(smem)[i] = 0u;
    __syncthreads();
  }
  float* coef = reinterpret_cast<float*>(smem + TILE_BYTES);  // [scale 32][shift 32] of this workgroup's channels
  const bool x_affine = X.scale != nullptr;
  if (x_affine) {
    if (tid < 32) {
      coef[tid] = tid < k_cnt ? X.scale[c0 + tid] : 1.f;
      coef[32 + tid] = tid < k_cnt ? X.shift[c0 + tid] : 0.f;
    }
    __syncthreads();
  }

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
  auto load_item = [&](int q) -> u32x4 {  // branch-free: clamped coordinates / channels, zeroing at the LDS write
    if (q < X_ITEMS) {
      const int it = tid + q * kWThreads;
      const int hp = min(it >> 2, NPIX - 1), cc = (it & 3) << 3;
      const int hy = hp / HWp, hx = hp - hy * HWp;
      const int y = min(max(ty0 + hy - HALO, 0), d.H - 1), x = min(max(tx0 + hx - HALO, 0), d.W - 1);
      return *reinterpret_cast<const u32x4*>(xptr + view_pixel_offset(X, img, y, x) + c0 + (cc < k_cnt ? cc : 0));
    }
    const int it = tid + (q - X_ITEMS) * kWThreads;
    const int p = it >> 2, cc = (it & 3) << 3;
    const int y = min(ty0 + (p >> LOG2TW), d.H - 1), x = min(tx0 + (p & (TW - 1)), d.W - 1);
    return *reinterpret_cast<const u32x4*>(dyptr + view_pixel_offset(DY, img, y, x) + nc0 + (cc < n_cnt ? cc : 0));
  };
  auto store_item = [&](int q, unsigned char* buf, u32x4 v) {
    if (q < X_ITEMS) {
      const int it = tid + q * kWThreads;
      const int hp = it >> 2, cc = (it & 3) << 3;
      const int hy = hp / HWp, hx = hp - hy * HWp;
      const int y = ty0 + hy - HALO, x = tx0 + hx - HALO;
      const bool keep = y >= 0 && y < d.H && x >= 0 && x < d.W;
      if (x_affine || X.relu) {
        float f[8];
        unpack8(v, f);
        if (x_affine) {
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], coef[cc + e], coef[32 + cc + e]);
        }
        if (X.relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
        }
        v = pack8(f);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = keep ? v[e] : 0u;
      if (it < NPIX * 4 && cc < k_cnt) *reinterpret_cast<u32x4*>(&buf[it * 16]) = v;
    } else {
      const int it = tid + (q - X_ITEMS) * kWThreads;
      const int p = it >> 2, cc = (it & 3) << 3;
      const bool keep = ty0 + (p >> LOG2TW) < d.H && tx0 + (p & (TW - 1)) < d.W;
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = keep ? v[e] : 0u;
      if (cc < n_cnt) *reinterpret_cast<u32x4*>(&buf[X_BYTES + it * 16]) = v;
    }
  };

  f32x16 acc[TAPS];
#pragma unroll
  for (int t = 0; t < TAPS; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float dbsum = 0.f;

  // transposed-read geometry of this lane: 16-lane group g16 -> channel block (g16 & 1), pixel half hh = g16 >> 1 (= h);
  // inside the group lane 4q + pp addresses pixel row q, channels 4pp..4pp+3 of the block
  const int cblock = (lane >> 4) & 1, q = (lane >> 2) & 3, pp = lane & 3;
  const int lane_off = cblock * 32 + pp * 8;
  // byte offsets of the lane's pixel in the x patch / dy tile for (step ks, block b): p = 32*wave + 16*ks + 8*h + 4*b + q
  unsigned xoff[2][2], yoff[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int p = 32 * wave + 16 * ks + 8 * h + 4 * b + q;
      xoff[ks][b] = static_cast<unsigned>(((p >> LOG2TW) * HWp + (p & (TW - 1))) * 64 + lane_off);
      yoff[ks][b] = static_cast<unsigned>(X_BYTES + p * 64 + lane_off);
    }
  const unsigned smem_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));

  auto compute = [&](unsigned buf_off) {
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      u32x2 yb[2], xa[TAPS][2];
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const unsigned ya = smem_base + buf_off + yoff[ks][b], xb = smem_base + buf_off + xoff[ks][b];
        lds_read_tr<0>(yb[b], ya);
        if constexpr (TAPS == 9) {
          lds_read_tr<(0 * HWp + 0) * 64>(xa[0][b], xb);
          lds_read_tr<(0 * HWp + 1) * 64>(xa[1 % TAPS][b], xb);
          lds_read_tr<(0 * HWp + 2) * 64>(xa[2 % TAPS][b], xb);
          lds_read_tr<(1 * HWp + 0) * 64>(xa[3 % TAPS][b], xb);
          lds_read_tr<(1 * HWp + 1) * 64>(xa[4 % TAPS][b], xb);
          lds_read_tr<(1 * HWp + 2) * 64>(xa[5 % TAPS][b], xb);
          lds_read_tr<(2 * HWp + 0) * 64>(xa[6 % TAPS][b], xb);
          lds_read_tr<(2 * HWp + 1) * 64>(xa[7 % TAPS][b], xb);
          lds_read_tr<(2 * HWp + 2) * 64>(xa[8 % TAPS][b], xb);
        } else {
          lds_read_tr<0>(xa[0][b], xb);
        }
      }
      lds_wait2(yb[0], yb[1]);
#pragma unroll
      for (int t = 0; t < TAPS; ++t) lds_wait2(xa[t][0], xa[t][1]);
      const u32x4 bfrag = {yb[0][0], yb[0][1], yb[1][0], yb[1][1]};
#pragma unroll
      for (int e = 0; e < 4; ++e) dbsum += bf_lo(bfrag[e]) + bf_hi(bfrag[e]);
#pragma unroll
      for (int t = 0; t < TAPS; ++t) {
        const u32x4 afrag = {xa[t][0][0], xa[t][0][1], xa[t][1][0], xa[t][1][1]};
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afrag), __builtin_bit_cast(bf16x8, bfrag),
                                                         acc[t], 0, 0, 0);
      }
    }
  };

  const long stride = gridDim.x;
  const long t0 = blockIdx.x;
  const long n_my = (t0 < a.n_pix_tiles) ? (a.n_pix_tiles - t0 + stride - 1) / stride : 0;
  if (n_my > 0) {
    set_tile(t0);
#pragma unroll
    for (int qi = 0; qi < N_ITEMS; ++qi) store_item(qi, smem, load_item(qi));
  }
  __syncthreads();
  for (long i = 0; i < n_my; ++i) {
    const unsigned cur = static_cast<unsigned>(i & 1) * BUF;
    unsigned char* nxt = smem + ((i + 1) & 1) * BUF;
    const bool more = i + 1 < n_my;
    u32x4 stage[N_ITEMS];
    if (more) {
      set_tile(t0 + (i + 1) * stride);
#pragma unroll
      for (int qi = 0; qi < N_ITEMS; ++qi) stage[qi] = load_item(qi);
    }
    compute(cur);
    if (more) {
#pragma unroll
      for (int qi = 0; qi < N_ITEMS; ++qi) store_item(qi, nxt, stage[qi]);
    }
    __syncthreads();
  }

  // ---- fixed-order tree sum of the 8 waves through LDS, then one slab per workgroup ----
  float* fs = reinterpret_cast<float*>(smem);
  tree_sum_waves<TAPS>(acc, fs, fs + 2 * TAPS * 1024, wave, lane);
  const long slab_stride = (static_cast<long>(TAPS) * a.Ktot + 1) * a.Ncols;
  float* slab = d.slabs + blockIdx.x * slab_stride;
  if (wave == 0) store_slab_block<TAPS>(acc, slab, a.Ktot, a.Ncols, kbase + c0, k_cnt, n0, n_cnt, j, h);
  if (want_db) {
    dbsum += __shfl_xor(dbsum, 32);
    float* dbs = fs + TAPS * 1024;
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

template <int TAPS, int LOG2TW>
int launch_one(const WBfArgs& a, dim3 grid, hipStream_t st) {
  constexpr int XPIX = (TAPS == 9) ? kMaxHaloPixels : kBlockPixels;
  constexpr int BUF = XPIX * 64 + kBlockPixels * 64;
  constexpr int TREE = 4 * TAPS * 4096;
  constexpr size_t lds = ((2 * BUF > TREE) ? 2 * BUF : TREE) + 256;
  if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_bf16_kernel<TAPS, LOG2TW>),
                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess)
    return UNETPP_ELAUNCH;
  hipLaunchKernelGGL((wgrad_bf16_kernel<TAPS, LOG2TW>), grid, dim3(kWThreads), lds, st, a);
  note_kernel(TAPS == 9 ? "wgrad_bf16_kernel<9>" : "wgrad_bf16_kernel<1>");
  return launch_status();
}

}  // namespace

// UNETPP_OK after launching, UNETPP_EINVAL when the views are not 8-channel aligned plain bf16 views (x may carry an
// affine + ReLU load transform; ReLU gates on load are not supported in bf16)
int launch_wgrad_bf16(const unetpp_wgrad_desc* d, int Ktot, int Ncols, int n_tiles_cols, int k_tiles, hipStream_t st) {
  for (int i = 0; i < d->n_x; ++i)
    if (!bf16_view_aligned(d->x[i]) || d->x[i].gate != nullptr) return UNETPP_EINVAL;
  for (int i = 0; i < d->n_dy; ++i)
    if (!bf16_view_aligned(d->dy[i]) || d->dy[i].gate != nullptr || d->dy[i].scale != nullptr || d->dy[i].relu)
      return UNETPP_EINVAL;
  WBfArgs a;
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

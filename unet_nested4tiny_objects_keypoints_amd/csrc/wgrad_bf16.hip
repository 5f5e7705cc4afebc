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
// Workgroup = 4 waves, TWO workgroups per CU: a wave owns 64 of the tile's 256 pixels (four 16-pixel MFMA steps) and
// keeps 9 taps x 16 accumulator registers over its whole pixel loop.  A 256-pixel tile is only 36 MFMAs (0.5 us) per
// wave, so the loop is paced by what surrounds them (loads, LDS stores, the barrier): with one 8-wave workgroup per CU
// all waves went through those phases in lockstep (~2 us per tile, tools history in DESIGN.md); two independent
// workgroups overlap one's staging and barrier with the other's MFMAs.  Staging: global -> registers -> LDS with the
// BatchNorm-apply + ReLU load transform of x in between (fp32, rounded to bf16); prefetch distance two with one register
// stage and two LDS buffers (at the top of tile i the registers hold tile i+1, requested a whole iteration ago; they go
// into the buffer tile i-1 was read from, tile i+2 is requested, then tile i's MFMAs run); one barrier per tile.
// (An LDS-DMA ring of three buffers was tried for plain views: same speed, the DMA instructions issue slowly.)
// Fixed-order tree sum of the 4 waves; one slab per workgroup in the format of wgrad.hip ([taps*K + 1][Ncols] fp32,
// last row = db).
#include <atomic>

#include "bf16_common.h"
#include "common.h"
#include "lds_asm.h"
#include "wgrad_reduce.h"

namespace unetpp {
namespace {

constexpr int kWThreads = 256;
constexpr int kWaves = 4;
constexpr int kSteps = 4;  // 16-pixel MFMA steps per wave and tile (64 pixels per wave)

struct WBfArgs {
  unetpp_wgrad_desc d;
  int tiles_x, tiles_y;
  int Ktot, Ncols, n_tiles_cols;
  long n_pix_tiles;
};

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef short s16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) s16x4* lds_tr_ptr;
// one transposed read (compiler builtin: the two halves of an MFMA operand land in adjacent registers and the byte
// offset folds into the instruction's immediate; as inline asm the operands had to be assembled with ~500 v_mov per
// tile loop body)
__device__ __forceinline__ s16x4 lds_tr(unsigned addr) {
  return __builtin_amdgcn_ds_read_tr16_b64_v4i16(reinterpret_cast<lds_tr_ptr>(static_cast<uintptr_t>(addr)));
}
// MFMA operand of a 16-pixel step: pixels +0..3 and +4..7 of the lane's half (two blocks 4 pixels = 256 bytes apart)
__device__ __forceinline__ s16x8 lds_tr_frag(unsigned addr) {
  return __builtin_shufflevector(lds_tr(addr), lds_tr(addr + 4 * 64), 0, 1, 2, 3, 4, 5, 6, 7);
}

// Fixed-order sum (w0 + w2) + (w1 + w3) of the four waves' accumulators through two LDS regions of TAPS*1024 floats
// (lane-linear [t][r/4][lane][r%4]: 16-byte accesses, no bank conflicts); the total lands in wave 0.
template <int TAPS>
__device__ __forceinline__ void tree_sum_4waves(f32x16 (&acc)[TAPS], float* regions, int wave, int lane) {
  constexpr int R = TAPS * 1024;
  auto put = [&](float* rg) {
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(rg + (t * 4 + q) * 256 + lane * 4) =
            f32x4{acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]};
  };
  auto add = [&](const float* rg) {
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(rg + (t * 4 + q) * 256 + lane * 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[t][4 * q + i] += v[i];
      }
  };
  if (wave >= 2) put(regions + (wave - 2) * R);
  __syncthreads();
  if (wave < 2) add(regions + wave * R);
  __syncthreads();
  if (wave == 1) put(regions);
  __syncthreads();
  if (wave == 0) add(regions);
}

template <int TAPS, int LOG2TW>
__global__ __launch_bounds__(kWThreads, 2) void wgrad_bf16_kernel(const WBfArgs a) {
  constexpr int HALO = (TAPS == 9) ? 1 : 0;
  constexpr int TW = 1 << LOG2TW, TH = kBlockPixels >> LOG2TW;
  constexpr int HWp = TW + 2 * HALO, HHp = TH + 2 * HALO;
  constexpr int NPIX = HWp * HHp;
  constexpr int XPIX = (TAPS == 9) ? kMaxHaloPixels : kBlockPixels;
  constexpr int X_ITEMS = (NPIX * 4 + kWThreads - 1) / kWThreads;  // 16-byte items (8 channels), 4 per pixel: <= 6
  constexpr int DY_ITEMS = (kBlockPixels * 4) / kWThreads;         // 4
  constexpr int N_ITEMS = X_ITEMS + DY_ITEMS;
  constexpr int X_BYTES = XPIX * 64;
  constexpr int DY_BYTES = kBlockPixels * 64;
  constexpr int BUF = X_BYTES + DY_BYTES;
  constexpr int TREE_BYTES = 2 * TAPS * 4096;                      // two regions of TAPS*1024 floats
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

  // ---- staging.  Per-thread constants: halo / tile coordinates of every item; interior tiles (all but the image
  // border) take their addresses as tile origin (uniform) + constant offset, without clamping. ----
  unsigned x_rel[X_ITEMS], y_rel[DY_ITEMS];
  auto item_hy = [&](int q) { return min((tid + q * kWThreads) >> 2, NPIX - 1) / HWp; };            // recomputed on the
  auto item_hx = [&](int q) { return min((tid + q * kWThreads) >> 2, NPIX - 1) - item_hy(q) * HWp; };  // border tiles only
  const int cc = (tid & 3) << 3;  // channel octet of every item of this thread (kWThreads % 4 == 0)
#pragma unroll
  for (int q = 0; q < X_ITEMS; ++q) {
    x_rel[q] = (static_cast<unsigned>(item_hy(q)) * X.sy * X.Ws + static_cast<unsigned>(item_hx(q)) * X.sx) * X.C + (cc < k_cnt ? cc : 0);
  }
#pragma unroll
  for (int q = 0; q < DY_ITEMS; ++q) {
    const int p = (tid + q * kWThreads) >> 2;
    y_rel[q] = (static_cast<unsigned>(p >> LOG2TW) * DY.sy * DY.Ws + static_cast<unsigned>(p & (TW - 1)) * DY.sx) * DY.C +
               (cc < n_cnt ? cc : 0);
  }
  int ty0 = 0, tx0 = 0, img = 0;  // tile being addressed
  auto set_tile = [&](long tile) {  // tile < 2^31 (launcher): 32-bit divisions
    unsigned b = static_cast<unsigned>(tile);
    const unsigned txi = b % static_cast<unsigned>(a.tiles_x);
    b /= static_cast<unsigned>(a.tiles_x);
    const unsigned tyi = b % static_cast<unsigned>(a.tiles_y);
    img = static_cast<int>(b / static_cast<unsigned>(a.tiles_y));
    ty0 = static_cast<int>(tyi) * TH;
    tx0 = static_cast<int>(txi) * TW;
  };
  auto is_interior = [&]() { return ty0 >= HALO && tx0 >= HALO && ty0 + TH + HALO <= d.H && tx0 + TW + HALO <= d.W; };
  auto load_tile = [&](u32x4 (&stage)[N_ITEMS]) {  // branch-free loads from valid addresses; zeroing at the LDS write
    if (is_interior()) {  // uniform
      const bf16_t* xo = xptr + view_pixel_offset(X, img, ty0 - HALO, tx0 - HALO) + c0;
      const bf16_t* yo = dyptr + view_pixel_offset(DY, img, ty0, tx0) + nc0;
#pragma unroll
      for (int q = 0; q < X_ITEMS; ++q) stage[q] = *reinterpret_cast<const u32x4*>(xo + x_rel[q]);
#pragma unroll
      for (int q = 0; q < DY_ITEMS; ++q) stage[X_ITEMS + q] = *reinterpret_cast<const u32x4*>(yo + y_rel[q]);
      return;
    }
#pragma unroll
    for (int q = 0; q < X_ITEMS; ++q) {
      const int y = min(max(ty0 + item_hy(q) - HALO, 0), d.H - 1), x = min(max(tx0 + item_hx(q) - HALO, 0), d.W - 1);
      stage[q] = *reinterpret_cast<const u32x4*>(xptr + view_pixel_offset(X, img, y, x) + c0 + (cc < k_cnt ? cc : 0));
    }
#pragma unroll
    for (int q = 0; q < DY_ITEMS; ++q) {
      const int p = (tid + q * kWThreads) >> 2;
      const int y = min(ty0 + (p >> LOG2TW), d.H - 1), x = min(tx0 + (p & (TW - 1)), d.W - 1);
      stage[X_ITEMS + q] = *reinterpret_cast<const u32x4*>(dyptr + view_pixel_offset(DY, img, y, x) + nc0 + (cc < n_cnt ? cc : 0));
    }
  };
  // writes the tile held in `stage` (geometry s_ty0 / s_tx0) into an LDS buffer
  auto store_tile = [&](const u32x4 (&stage)[N_ITEMS], int s_ty0, int s_tx0, unsigned char* buf) {
    const bool interior = s_ty0 >= HALO && s_tx0 >= HALO && s_ty0 + TH + HALO <= d.H && s_tx0 + TW + HALO <= d.W;
#pragma unroll
    for (int q = 0; q < X_ITEMS; ++q) {
      const int it = tid + q * kWThreads;
      u32x4 v = stage[q];
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
      if (!interior) {
        const int y = s_ty0 + item_hy(q) - HALO, x = s_tx0 + item_hx(q) - HALO;
        const bool keep = y >= 0 && y < d.H && x >= 0 && x < d.W;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = keep ? v[e] : 0u;
      }
      if (it < NPIX * 4 && cc < k_cnt) *reinterpret_cast<u32x4*>(&buf[it * 16]) = v;
    }
#pragma unroll
    for (int q = 0; q < DY_ITEMS; ++q) {
      const int it = tid + q * kWThreads;
      u32x4 v = stage[X_ITEMS + q];
      if (!interior) {
        const int p = it >> 2;
        const bool keep = s_ty0 + (p >> LOG2TW) < d.H && s_tx0 + (p & (TW - 1)) < d.W;
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = keep ? v[e] : 0u;
      }
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
  const int cblock = (lane >> 4) & 1, q4 = (lane >> 2) & 3, pp = lane & 3;
  const int lane_off = cblock * 32 + pp * 8;
  // The lane's pixel for (step ks, block b) is p = 64*wave + 16*ks + 8*h + 4*b + q4.  With the patch shape a template
  // parameter, (ks, b) only add whole rows and a column shift that never carries (TW = 32: row 2w + (ks >> 1), column
  // 16*(ks & 1) + 8h + 4b + q4 <= 31; TW = 16: row 4w + ks, column 8h + 4b + q4; TW = 8: row 8w + 2ks + h, column
  // 4b + q4), so ONE base address per operand and immediate offsets address every read of the tile loop.
  const int p_lane = 64 * wave + 8 * h + q4;
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
  const unsigned x_base = lds0 + static_cast<unsigned>(((p_lane >> LOG2TW) * HWp + (p_lane & (TW - 1))) * 64 + lane_off);
  const unsigned y_base = lds0 + static_cast<unsigned>(X_BYTES + p_lane * 64 + lane_off);

  // one MFMA per (step, tap)
  auto compute = [&](unsigned buf_off) {
    const unsigned xb = x_base + buf_off, yb_addr = y_base + buf_off;
    static_for<kSteps>([&](auto kc) {
      constexpr int ks = decltype(kc)::v;
      constexpr int DROW = LOG2TW == 5 ? (ks >> 1) : (LOG2TW == 4 ? ks : 2 * ks);
      constexpr int DCOL = LOG2TW == 5 ? 16 * (ks & 1) : 0;
      constexpr int X0 = (DROW * HWp + DCOL) * 64;
      constexpr int Y0 = (16 * ks) * 64;
      const s16x8 bfrag = lds_tr_frag(yb_addr + Y0);
      const u32x4 bu = __builtin_bit_cast(u32x4, bfrag);
#pragma unroll
      for (int e = 0; e < 4; ++e) dbsum += bf_lo(bu[e]) + bf_hi(bu[e]);
      static_for<TAPS>([&](auto tc) {
        constexpr int t = decltype(tc)::v;
        constexpr int TOFF = (TAPS == 9) ? ((t / 3) * HWp + (t % 3)) * 64 : 0;
        const s16x8 afrag = lds_tr_frag(xb + X0 + TOFF);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afrag), __builtin_bit_cast(bf16x8, bfrag),
                                                         acc[t], 0, 0, 0);
        // a scheduling fence per filter row: left alone hipcc hoists all 72 reads of a tile to the top of the loop body,
        // and with 144 accumulators and 40 staging registers live that spills
        if constexpr (t % 3 == 2) __builtin_amdgcn_sched_barrier(0);
      });
    });
  };

  // ---- tile loop: prefetch distance two (one register stage, two LDS buffers), one barrier per tile ----
  const long stride = gridDim.x;
  const long t0 = blockIdx.x;
  const long n_my = (t0 < a.n_pix_tiles) ? (a.n_pix_tiles - t0 + stride - 1) / stride : 0;
  u32x4 stage[N_ITEMS];
  int s_ty0 = 0, s_tx0 = 0;  // geometry of the tile held in `stage`
  if (n_my > 0) {
    set_tile(t0);
    load_tile(stage);
    store_tile(stage, ty0, tx0, smem);
  }
  if (n_my > 1) {
    set_tile(t0 + stride);
    s_ty0 = ty0;
    s_tx0 = tx0;
    load_tile(stage);
  }
  __syncthreads();
  for (long i = 0; i < n_my; ++i) {
    const unsigned cur = static_cast<unsigned>(i & 1) * BUF;
    if (i + 1 < n_my) store_tile(stage, s_ty0, s_tx0, smem + ((i + 1) & 1) * BUF);  // buffer last read a barrier ago
    if (i + 2 < n_my) {
      set_tile(t0 + (i + 2) * stride);
      s_ty0 = ty0;
      s_tx0 = tx0;
      load_tile(stage);
    }
    compute(cur);
    __syncthreads();
  }

  // ---- fixed-order tree sum of the 4 waves through LDS, then one slab per workgroup ----
  float* fs = reinterpret_cast<float*>(smem);
  tree_sum_4waves<TAPS>(acc, fs, wave, lane);
  const long slab_stride = (static_cast<long>(TAPS) * a.Ktot + 1) * a.Ncols;
  float* slab = d.slabs + blockIdx.x * slab_stride;
  if (wave == 0) store_slab_block<TAPS>(acc, slab, a.Ktot, a.Ncols, kbase + c0, k_cnt, n0, n_cnt, j, h);
  if (want_db) {
    dbsum += __shfl_xor(dbsum, 32);
    float* dbs = fs + TAPS * 1024;  // second region: dead after the tree's last round
    __syncthreads();
    if (h == 0) dbs[wave * 32 + j] = dbsum;
    __syncthreads();
    if (tid < n_cnt) {
      float sum = 0.f;
#pragma unroll
      for (int w = 0; w < kWaves; ++w) sum += dbs[w * 32 + tid];
      slab[static_cast<long>(TAPS) * a.Ktot * a.Ncols + n0 + tid] = sum;
    }
  }
}

template <int TAPS, int LOG2TW>
int launch_one(const WBfArgs& a, dim3 grid, hipStream_t st) {
  constexpr int XPIX = (TAPS == 9) ? kMaxHaloPixels : kBlockPixels;
  constexpr int BUF = XPIX * 64 + kBlockPixels * 64;
  constexpr int TREE = 2 * TAPS * 4096;
  constexpr size_t lds = ((2 * BUF > TREE) ? 2 * BUF : TREE) + 256;
  static_assert(2 * lds <= 160 * 1024, "two workgroups per CU");
  // > 64 KB of dynamic LDS needs the per-function opt-in; the attribute is PER DEVICE, so it is remembered per device
  // (bit mask, atomically updated: host threads may launch concurrently) -- the call costs tens of microseconds of
  // host time, more than a short launch runs on the device, hence not on every launch
  static std::atomic<unsigned long long> opted_in[4] = {};  // devices 0..255
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess || device < 0 || device > 255) return UNETPP_ELAUNCH;
  const unsigned long long bit = 1ull << (device & 63);
  if (!(opted_in[device >> 6].load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_bf16_kernel<TAPS, LOG2TW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess)
      return UNETPP_ELAUNCH;
    opted_in[device >> 6].fetch_or(bit, std::memory_order_release);
  }
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
  if (a.n_pix_tiles > 0x7fffffffL) return UNETPP_EINVAL;
  for (int i = 0; i < d->n_x; ++i)   // 32-bit element offsets inside a tile
    if (static_cast<long>(d->x[i].Hs) * d->x[i].Ws * d->x[i].C >= 0x7fffffffL) return UNETPP_EINVAL;
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

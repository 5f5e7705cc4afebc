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
#include <cstdlib>

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
constexpr unsigned kOutOfRange = 0x80000000u;  // buffer offset no image reaches (images are below 2 GB: launcher)
// buffer resource over ONE image of a bf16 NHWC tensor
__device__ __forceinline__ __amdgpu_buffer_rsrc_t image_rsrc(const bf16_t* base, int img, int img_elems) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(base + static_cast<long>(img) * img_elems), 0, img_elems * 2,
                                           0x00020000);
}

typedef unsigned u32x4v __attribute__((vector_size(16)));  // the buffer-load builtin's own return type
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

  // ---- staging.  Both operands come through buffer resources (one per image, rebuilt in scalar registers per tile):
  // an item's byte offset is the tile origin (scalar; "negative" for the halo of a border tile, in wrapping 32-bit
  // arithmetic) + a per-thread constant, and an item outside the image -- or a channel octet past a narrow view -- gets
  // an offset past the end of the image: the hardware returns zeros without touching memory.  One path for interior and
  // border tiles (the clamped-address border path cost ~100 vector instructions per item). ----
  const int cc = (tid & 3) << 3;  // channel octet of every item of this thread (kWThreads % 4 == 0)
  unsigned x_rel[X_ITEMS], y_rel[DY_ITEMS];
  int x_pos[X_ITEMS];   // (halo row << 16) | halo column of the item; rows past the patch never pass the range test
#pragma unroll
  for (int q = 0; q < X_ITEMS; ++q) {
    const int hp = (tid + q * kWThreads) >> 2;
    const int hy = hp / HWp, hx = hp - hy * HWp;
    x_pos[q] = (hp < NPIX && cc < k_cnt) ? ((hy << 16) | hx) : (0x4000 << 16);
    x_rel[q] = ((static_cast<unsigned>(hy) * X.sy * X.Ws + static_cast<unsigned>(hx) * X.sx) * X.C + cc) * 2u;
  }
#pragma unroll
  for (int q = 0; q < DY_ITEMS; ++q) {
    const int p = (tid + q * kWThreads) >> 2;
    y_rel[q] = ((static_cast<unsigned>(p >> LOG2TW) * DY.sy * DY.Ws + static_cast<unsigned>(p & (TW - 1)) * DY.sx) * DY.C + cc) * 2u;
  }
  const int x_img_elems = X.Hs * X.Ws * X.C, y_img_elems = DY.Hs * DY.Ws * DY.C;
  auto x_inside = [&](int q, int y0, int x0) __attribute__((always_inline)) {
    const int y = y0 - HALO + (x_pos[q] >> 16), x = x0 - HALO + (x_pos[q] & 0xffff);
    return (static_cast<unsigned>(y) < static_cast<unsigned>(d.H)) & (static_cast<unsigned>(x) < static_cast<unsigned>(d.W));
  };
  auto y_inside = [&](int q, int y0, int x0) __attribute__((always_inline)) {
    const int p = (tid + q * kWThreads) >> 2;
    return (cc < n_cnt) & (y0 + (p >> LOG2TW) < d.H) & (x0 + (p & (TW - 1)) < d.W);
  };
  int ty0 = 0, tx0 = 0, img = 0;  // tile being addressed
  auto set_tile = [&](long tile) __attribute__((always_inline)) {  // tile < 2^31 (launcher): 32-bit divisions
    unsigned b = static_cast<unsigned>(tile);
    const unsigned txi = b % static_cast<unsigned>(a.tiles_x);
    b /= static_cast<unsigned>(a.tiles_x);
    const unsigned tyi = b % static_cast<unsigned>(a.tiles_y);
    img = static_cast<int>(b / static_cast<unsigned>(a.tiles_y));
    ty0 = static_cast<int>(tyi) * TH;
    tx0 = static_cast<int>(txi) * TW;
  };
  auto load_tile = [&](u32x4 (&stage)[N_ITEMS]) __attribute__((always_inline)) {
    const __amdgpu_buffer_rsrc_t xr = image_rsrc(xptr, img, x_img_elems), yr = image_rsrc(dyptr, img, y_img_elems);
    const unsigned xorg =
        static_cast<unsigned>((((ty0 - HALO) * X.sy + X.oy) * X.Ws + (tx0 - HALO) * X.sx + X.ox) * X.C + X.c_off + c0) * 2u;
    const unsigned yorg = static_cast<unsigned>(((ty0 * DY.sy + DY.oy) * DY.Ws + tx0 * DY.sx + DY.ox) * DY.C + DY.c_off + nc0) * 2u;
#pragma unroll
    for (int q = 0; q < X_ITEMS; ++q) {
      const unsigned off = x_inside(q, ty0, tx0) ? xorg + x_rel[q] : kOutOfRange;
      stage[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(xr, static_cast<int>(off), 0, 0));
    }
#pragma unroll
    for (int q = 0; q < DY_ITEMS; ++q) {
      const unsigned off = y_inside(q, ty0, tx0) ? yorg + y_rel[q] : kOutOfRange;
      stage[X_ITEMS + q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(yr, static_cast<int>(off), 0, 0));
    }
  };
  // writes the tile held in `stage` (geometry s_ty0 / s_tx0) into an LDS buffer
  auto store_tile = [&](const u32x4 (&stage)[N_ITEMS], int s_ty0, int s_tx0, unsigned char* buf) __attribute__((always_inline)) {
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
        if (x_affine) {   // the padding is zero AFTER the transform
          const bool keep = x_inside(q, s_ty0, s_tx0);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = keep ? v[e] : 0u;
        }
      }
      if (it < NPIX * 4) *reinterpret_cast<u32x4*>(&buf[it * 16]) = v;
    }
#pragma unroll
    for (int q = 0; q < DY_ITEMS; ++q) {
      const int it = tid + q * kWThreads;
      *reinterpret_cast<u32x4*>(&buf[X_BYTES + it * 16]) = stage[X_ITEMS + q];
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

  // One MFMA per (step, tap).  A filter row only moves the x fragment by one image row, so the wave walks the INPUT
  // rows of its 64 pixels: the fragment of input row i (one per filter column and 16-pixel half) feeds the output rows
  // i, i-1, i-2 under filter rows 0, 1, 2 against the four dy fragments of the tile, which stay in registers --
  // 24 x + 4 dy fragment reads per 36 MFMAs where reading per (step, tap) took 72 + 8: 4.4 ds_read_b64_tr_b16 per
  // MFMA kept the LDS array busy 36 cycles of every 32-cycle MFMA slot of the CU's four SIMDs.
  auto compute = [&](unsigned buf_off) {
    const unsigned xb = x_base + buf_off, yb_addr = y_base + buf_off;
    s16x8 bfr[kSteps];
    static_for<kSteps>([&](auto kc) {
      constexpr int ks = decltype(kc)::v;
      bfr[ks] = lds_tr_frag(yb_addr + 16 * ks * 64);
      const u32x4 bu = __builtin_bit_cast(u32x4, bfr[ks]);
#pragma unroll
      for (int e = 0; e < 4; ++e) dbsum += bf_lo(bu[e]) + bf_hi(bu[e]);
    });
    if constexpr (TAPS == 9) {
      constexpr int NCOL = LOG2TW == 5 ? 2 : 1;          // 16-pixel steps per image row
      constexpr int RSTEP = LOG2TW == 3 ? 2 : 1;         // image rows per step
      constexpr int ROWS = LOG2TW == 5 ? 2 : (LOG2TW == 4 ? 4 : 8);   // image rows of the wave's 64 pixels
      static_for<ROWS + 2>([&](auto kk) {
        constexpr int key = decltype(kk)::v;             // input row (halo included)
        constexpr bool used = [] {
          for (int r = 0; r < 3; ++r) {
            const int ro = key - r;
            if (ro >= 0 && ro < ROWS && ro % RSTEP == 0) return true;
          }
          return false;
        }();
        if constexpr (used) {
          static_for<NCOL>([&](auto cv) {
            constexpr int c = decltype(cv)::v;
            static_for<3>([&](auto sv) {
              constexpr int sh = decltype(sv)::v;        // filter column
              const s16x8 afrag = lds_tr_frag(xb + (key * HWp + 16 * c + sh) * 64);
              static_for<3>([&](auto rv) {
                constexpr int r = 2 - decltype(rv)::v;   // filter row
                constexpr int ro = key - r;
                if constexpr (ro >= 0 && ro < ROWS && ro % RSTEP == 0) {
                  constexpr int ks = (ro / RSTEP) * NCOL + c;
                  acc[3 * r + sh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                      __builtin_bit_cast(bf16x8, afrag), __builtin_bit_cast(bf16x8, bfr[ks]), acc[3 * r + sh], 0, 0, 0);
                }
              });
            });
            // a scheduling fence per input row half: left alone hipcc hoists all reads of a tile to the top of the
            // loop body, and with 144 accumulators and 40 staging registers live that spills
            __builtin_amdgcn_sched_barrier(0);
          });
        }
      });
    } else {
      static_for<kSteps>([&](auto kc) {
        constexpr int ks = decltype(kc)::v;
        constexpr int DROW = LOG2TW == 5 ? (ks >> 1) : (LOG2TW == 4 ? ks : 2 * ks);
        constexpr int DCOL = LOG2TW == 5 ? 16 * (ks & 1) : 0;
        const s16x8 afrag = lds_tr_frag(xb + (DROW * HWp + DCOL) * 64);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afrag), __builtin_bit_cast(bf16x8, bfr[ks]),
                                                         acc[0], 0, 0, 0);
      });
    }
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


#ifdef UNETPP_WQ_EXP_CLOCK
// diagnostic build (tools/wgrad_quad_ablation.sh CLOCK): shader clocks and 100 MHz reference ticks spent in the tile
// loops, summed over workgroups -- in-kernel clock = cycles / ticks * 100 MHz (MI355X_MICROARCH.md, DVFS item 6)
__device__ unsigned long long g_wq_clock[2];
#endif

// ---------------------------------------------------------------------------------------------------------------------
// Wide layers -- every x and dy view a multiple of 64 channels.  The kernel above gives a workgroup ONE (32-channel,
// 32-column) pair and every MFMA its own operand reads: (18 x + 2 dy) fragments per 9 MFMAs, 4.4 ds_read_b64_tr_b16
// per MFMA -- with four SIMDs issuing, 36 LDS-array cycles per 32-cycle MFMA slot: the LDS array, not the matrix pipe,
// paces it (0.33 matrix-pipe occupancy in the PMC pass), and its global loads are 64-byte slices of pixel rows (half
// an L2 line), each x slice fetched again by every column tile and each dy slice by every channel tile.
// Here a workgroup owns a QUAD, 64 channels x 64 columns = 2 x 2 pairs, and stages 128-byte rows (whole L2 lines, half
// the bytes per pair).  3x3: THREE waves per pair, wave s owning the filter COLUMN s (taps s, 3+s, 6+s: 48
// accumulator registers) over all 256 pixels of the tile.  A filter row only shifts the x fragment by one image row,
// so the wave walks the INPUT rows of the patch: the fragment of input row i feeds output rows i, i-1, i-2 (filter
// rows 0, 1, 2) against dy fragments it keeps for three rows -- 20 x + 16 dy fragments per 48 MFMAs (1.5 reads per
// MFMA), no sum across waves at the end, every wave writes its three taps of the slab block.  12 waves, ONE workgroup
// per CU (two 75 KB LDS buffers).  1x1 (transposed-convolution phases): two waves per pair, half the pixels each,
// summed through LDS.
template <int TAPS>
struct QuadShape {
  static constexpr int WPP = TAPS == 9 ? 3 : 2;  // waves per pair
  static constexpr int THREADS = 4 * WPP * 64;
};

template <int TAPS, int LOG2TW>
__global__ __launch_bounds__(QuadShape<TAPS>::THREADS, 1) void wgrad_bf16_quad_kernel(const WBfArgs a) {
  constexpr int WPP = QuadShape<TAPS>::WPP, THREADS = QuadShape<TAPS>::THREADS;
  constexpr int HALO = (TAPS == 9) ? 1 : 0;
  constexpr int TW = 1 << LOG2TW, TH = kBlockPixels >> LOG2TW;
  constexpr int HWp = TW + 2 * HALO, HHp = TH + 2 * HALO;
  constexpr int NPIX = HWp * HHp;
  constexpr int XPIX = (TAPS == 9) ? kMaxHaloPixels : kBlockPixels;
  constexpr int X_ITEMS = (NPIX * 8 + THREADS - 1) / THREADS;          // 16-byte items (8 channels), 8 per pixel
  constexpr int DY_ITEMS = (kBlockPixels * 8 + THREADS - 1) / THREADS;
  constexpr int N_ITEMS = X_ITEMS + DY_ITEMS;                          // 7 (3x3) / 8 (1x1)
  // a plane = one 32-channel tile of all pixels ([pixel][64 bytes], the layout the transposed reads want); the two
  // planes of an operand are 128 bytes out of phase so that a pixel's two 64-byte halves fall into different banks
  constexpr int XP = XPIX * 64 + 128;
  constexpr int YP = kBlockPixels * 64 + 128;
  constexpr int BUF = 2 * XP + 2 * YP;
  static_assert(TAPS == 9 || 4 * TAPS * 4096 + 512 <= 2 * BUF, "the pair sums alias the tile buffers");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];  // 2 * BUF + 512

  const unetpp_wgrad_desc& d = a.d;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, j = lane & 31, h = lane >> 5;  // (scalar wave index)
  const int pair = wave / WPP, sub = wave - pair * WPP;  // sub = filter column (3x3) / pixel half (1x1)
  const int kl = pair >> 1, nl = pair & 1;               // (channel tile, column tile) of the pair inside the quad

  int nq = blockIdx.y % a.n_tiles_cols;   // n_tiles_cols = column QUADS here
  int kq = blockIdx.y / a.n_tiles_cols;
  const bool want_db = kq == 0;
  int dv = 0, col_base = 0;
  while (dv < d.n_dy - 1) {
    const int quads_v = d.dy[dv].c_len >> 6;
    if (nq < quads_v) break;
    nq -= quads_v;
    col_base += d.dy[dv].c_len;
    ++dv;
  }
  const unetpp_view& DY = d.dy[dv];
  int xv = 0, kbase = 0;
  while (xv < d.n_x - 1) {
    const int quads_v = d.x[xv].c_len >> 6;
    if (kq < quads_v) break;
    kq -= quads_v;
    kbase += d.x[xv].c_len;
    ++xv;
  }
  const unetpp_view& X = d.x[xv];
  const bf16_t* xptr = reinterpret_cast<const bf16_t*>(X.ptr);
  const bf16_t* dyptr = reinterpret_cast<const bf16_t*>(DY.ptr);
  const int c0 = kq * 64, nc0 = nq * 64;

  float* coef = reinterpret_cast<float*>(smem + 2 * BUF);  // [scale 64][shift 64] of this workgroup's channels
  const bool x_affine = X.scale != nullptr;
  if (x_affine) {
    if (tid < 64) {
      coef[tid] = X.scale[c0 + tid];
      coef[64 + tid] = X.shift[c0 + tid];
    }
    __syncthreads();
  }

  // ---- staging: item = (pixel, channel octet 0..7); a thread keeps its octet (THREADS % 8 == 0), eight neighbouring
  // threads fetch the 128 contiguous bytes of a pixel ----
  const int oct = tid & 7;
  const int ch = oct << 3;                                   // channel inside the 64
  const unsigned plane = static_cast<unsigned>(oct >> 2);    // 32-channel plane of the octet
  const unsigned x_plane = plane * XP + (oct & 3) * 16;
  const unsigned y_plane = 2u * XP + plane * YP + (oct & 3) * 16;
  // Both operands come through buffer resources: an item's byte offset is the tile origin (scalar; "negative" for
  // the halo of a border tile, in wrapping 32-bit arithmetic) + a per-thread constant, and an item outside the image
  // gets an offset past the end of the tensor -- the hardware returns zeros without touching memory.  One path for
  // interior and border tiles (the clamped-address border path cost ~100 vector instructions per item, and 30 % of the
  // tiles of a 256 x 256 image are border tiles).
  unsigned x_rel[X_ITEMS], y_rel[DY_ITEMS];
  int x_pos[X_ITEMS];   // (halo row << 16) | halo column of the item; rows past the patch never pass the range test
#pragma unroll
  for (int q = 0; q < X_ITEMS; ++q) {
    const int hp = (tid + q * THREADS) >> 3;
    const int hy = hp / HWp, hx = hp - hy * HWp;
    x_pos[q] = hp < NPIX ? ((hy << 16) | hx) : (0x4000 << 16);
    x_rel[q] = ((static_cast<unsigned>(hy) * X.sy * X.Ws + static_cast<unsigned>(hx) * X.sx) * X.C + ch) * 2u;
  }
#pragma unroll
  for (int q = 0; q < DY_ITEMS; ++q) {
    const int p = (tid + q * THREADS) >> 3;
    y_rel[q] = ((static_cast<unsigned>(p >> LOG2TW) * DY.sy * DY.Ws + static_cast<unsigned>(p & (TW - 1)) * DY.sx) * DY.C + ch) * 2u;
  }
  // one resource per IMAGE (base = the image's first element, rebuilt in scalar registers per tile): offsets stay
  // below 2^31 whatever the batch size
  const int x_img_elems = X.Hs * X.Ws * X.C, y_img_elems = DY.Hs * DY.Ws * DY.C;
  // range tests of an item of the tile at (y0, x0)
  auto x_inside = [&](int q, int y0, int x0) __attribute__((always_inline)) {
    const int y = y0 - HALO + (x_pos[q] >> 16), x = x0 - HALO + (x_pos[q] & 0xffff);
    return (static_cast<unsigned>(y) < static_cast<unsigned>(d.H)) & (static_cast<unsigned>(x) < static_cast<unsigned>(d.W));
  };
  auto y_inside = [&](int q, int y0, int x0) __attribute__((always_inline)) {
    const int p = (tid + q * THREADS) >> 3;
    return (p < kBlockPixels) & (y0 + (p >> LOG2TW) < d.H) & (x0 + (p & (TW - 1)) < d.W);
  };
  int ty0 = 0, tx0 = 0, img = 0;
  auto set_tile = [&](long tile) __attribute__((always_inline)) {
    unsigned b = static_cast<unsigned>(tile);
    const unsigned txi = b % static_cast<unsigned>(a.tiles_x);
    b /= static_cast<unsigned>(a.tiles_x);
    const unsigned tyi = b % static_cast<unsigned>(a.tiles_y);
    img = static_cast<int>(b / static_cast<unsigned>(a.tiles_y));
    ty0 = static_cast<int>(tyi) * TH;
    tx0 = static_cast<int>(txi) * TW;
  };
  // one staged item (q < X_ITEMS: an x pixel octet, else a dy pixel octet) at a time, so that the tile loop can place
  // the items between its MFMAs
  u32x4 stage[N_ITEMS];
  auto load_item = [&](auto qc) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::v;
    if constexpr (q < X_ITEMS) {
      const __amdgpu_buffer_rsrc_t rsrc = image_rsrc(xptr, img, x_img_elems);
      const unsigned org =
          static_cast<unsigned>((((ty0 - HALO) * X.sy + X.oy) * X.Ws + (tx0 - HALO) * X.sx + X.ox) * X.C + X.c_off + c0) * 2u;
      const unsigned off = x_inside(q, ty0, tx0) ? org + x_rel[q] : kOutOfRange;
      stage[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, static_cast<int>(off), 0, 0));
    } else {
      constexpr int qy = q - X_ITEMS;
      const __amdgpu_buffer_rsrc_t rsrc = image_rsrc(dyptr, img, y_img_elems);
      const unsigned org = static_cast<unsigned>(((ty0 * DY.sy + DY.oy) * DY.Ws + tx0 * DY.sx + DY.ox) * DY.C + DY.c_off + nc0) * 2u;
      const unsigned off = y_inside(qy, ty0, tx0) ? org + y_rel[qy] : kOutOfRange;
      stage[q] = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc, static_cast<int>(off), 0, 0));
    }
  };
  // bias gradient = column sums of dy: taken where every dy value passes through a thread's registers exactly once
  // (the staging writes), not beside the MFMAs -- there 16 VALU instructions per dy fragment filled the SIMD's vector
  // issue (4 cycles each against 24 free cycles per MFMA) and the loop ran at half the matrix rate
  float dbacc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  // writes item q of the tile held in `stage` (geometry s_y0 / s_x0) into an LDS buffer
  auto store_item = [&](auto qc, int s_y0, int s_x0, unsigned char* buf) __attribute__((always_inline)) {
    constexpr int q = decltype(qc)::v;
    u32x4 v = stage[q];
    if constexpr (q < X_ITEMS) {
      const int hp = (tid + q * THREADS) >> 3;
      if (x_affine || X.relu) {
        float f[8];
        unpack8(v, f);
        if (x_affine) {
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], coef[ch + e], coef[64 + ch + e]);
        }
        if (X.relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
        }
        v = pack8(f);
        if (x_affine) {   // the padding is zero AFTER the transform
          const bool keep = x_inside(q, s_y0, s_x0);
#pragma unroll
          for (int e = 0; e < 4; ++e) v[e] = keep ? v[e] : 0u;
        }
      }
      if (hp < NPIX) *reinterpret_cast<u32x4*>(&buf[x_plane + hp * 64]) = v;
    } else {
      constexpr int qy = q - X_ITEMS;
      const int p = (tid + qy * THREADS) >> 3;
      if (p < kBlockPixels) {
        *reinterpret_cast<u32x4*>(&buf[y_plane + p * 64]) = v;
        if (want_db) {   // uniform; the thread's 8 columns of this pixel
          float f[8];
          unpack8(v, f);
#pragma unroll
          for (int e = 0; e < 8; ++e) dbacc[e] += f[e];
        }
      }
    }
  };

  constexpr int NACC = TAPS == 9 ? 3 : 1;   // 3x3: acc[r] = tap (filter row r, filter column sub)
  f32x16 acc[NACC];
#pragma unroll
  for (int t = 0; t < NACC; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // transposed-read geometry as in the kernel above; the lane's pixel for (step ks, block b) is
  // p = p0 + 16*ks + 8*h + 4*b + q4 with p0 = 0 (3x3: all 16 steps) or 128 * sub (1x1: 8 steps), i.e. step ks sits at
  // image row DROW(ks), column DCOL(ks) of the patch (TW = 32: ks >> 1, 16*(ks & 1); TW = 16: ks, 0; TW = 8: 2ks, 0 --
  // there the lane's h is the second row)
  const int cblock = (lane >> 4) & 1, q4 = (lane >> 2) & 3, pp = lane & 3;
  const int lane_off = cblock * 32 + pp * 8;
  const int p_lane = (TAPS == 9 ? 0 : 128 * sub) + 8 * h + q4;
  const int x_shift = TAPS == 9 ? sub : 0;   // filter column
  const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
  const unsigned x_base =
      lds0 + static_cast<unsigned>(kl * XP + ((p_lane >> LOG2TW) * HWp + (p_lane & (TW - 1)) + x_shift) * 64 + lane_off);
  const unsigned y_base = lds0 + static_cast<unsigned>(2 * XP + nl * YP + p_lane * 64 + lane_off);

  auto compute = [&](unsigned buf_off, auto&& hook) __attribute__((always_inline)) {
    const unsigned xb = x_base + buf_off, yb_addr = y_base + buf_off;
    if constexpr (TAPS == 9) {
      constexpr int NCOL = LOG2TW == 5 ? 2 : 1;          // 16-pixel steps per image row
      constexpr int RSTEP = LOG2TW == 3 ? 2 : 1;         // image rows per step
      constexpr int ROWS = LOG2TW == 5 ? 8 : (LOG2TW == 4 ? 16 : 32);
      // fragments of input row `key` (x) / output row `key` (dy) are requested one row ahead of the MFMAs that use
      // them; the step at output row ro = key - r meets input row `key` under filter row r.  TW = 32: the two 16-pixel
      // halves of the rows one after the other (half the live fragments: two x and four dy).
      auto row_used = [](int key) {
        for (int r = 0; r < 3; ++r) {
          const int ro = key - r;
          if (ro >= 0 && ro < ROWS && ro % RSTEP == 0) return true;
        }
        return false;
      };
      static_for<NCOL>([&](auto cv) {
        constexpr int c = decltype(cv)::v;
        s16x8 afr[ROWS + 2], bfr[ROWS];
        auto request = [&](auto kk) __attribute__((always_inline)) {
          constexpr int key = decltype(kk)::v;
          if constexpr (key < ROWS + 2 && row_used(key)) {
            afr[key] = lds_tr_frag(xb + (key * HWp + 16 * c) * 64);
            if constexpr (key < ROWS && key % RSTEP == 0)   // a new output row: its dy fragment
              bfr[key] = lds_tr_frag(yb_addr + 16 * ((key / RSTEP) * NCOL + c) * 64);
          }
        };
        request(IC<0>{});
        static_for<ROWS + 2>([&](auto kk) {
          constexpr int key = decltype(kk)::v;
          request(IC<key + 1>{});
          static_for<3>([&](auto rv) {
            constexpr int r = 2 - decltype(rv)::v;          // oldest dy fragment first
            constexpr int ro = key - r;
            if constexpr (ro >= 0 && ro < ROWS && ro % RSTEP == 0)
              acc[r] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afr[key]),
                                                               __builtin_bit_cast(bf16x8, bfr[ro]), acc[r], 0, 0, 0);
          });
          hook(IC<c * (ROWS + 2) + key>{});   // one piece of the staging beside this row's MFMAs
#ifndef UNETPP_WQ_EXP_NO_FENCE
          __builtin_amdgcn_sched_barrier(0);   // keep the reads of later rows out of this row's registers
#endif
        });
      });
    } else {
      static_for<8>([&](auto kc) {
        constexpr int ks = decltype(kc)::v;
        constexpr int DROW = LOG2TW == 5 ? (ks >> 1) : (LOG2TW == 4 ? ks : 2 * ks);
        constexpr int DCOL = LOG2TW == 5 ? 16 * (ks & 1) : 0;
        const s16x8 bfrag = lds_tr_frag(yb_addr + 16 * ks * 64);
        const s16x8 afrag = lds_tr_frag(xb + (DROW * HWp + DCOL) * 64);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, afrag), __builtin_bit_cast(bf16x8, bfrag),
                                                         acc[0], 0, 0, 0);
        hook(IC<2 * ks>{});
        hook(IC<2 * ks + 1>{});
      });
    }
  };

  // ---- tile loop: the registers hold tile i+1 (requested an iteration ago): all waves write it into the other LDS
  // buffer and request tile i+2, then all run tile i's MFMAs; one barrier per tile.  In cycles a tile is 4.6 k of MFMAs
  // per SIMD + 1.6 k of gaps around them + 2.5 k of staging in front.  Two ways of hiding the staging were built and
  // measured on [64,64,64] -> 64 at 256 x 256 x 8 (wgrad + finish, us): this order 136-148; the first wave of every
  // SIMD staging while the other two multiply 151; one staged item beside each MFMA row (compute(cur, hook) below)
  // 169 -- more cycles per tile (the waits of the staging stream sit in front of MFMAs of the same wave), although the
  // chip then holds 2.32 GHz instead of 1.93 (random operands; 2.38 on zeros: tools/wgrad_quad_clock.py) ----
  const long stride = gridDim.x;
  const long t0 = blockIdx.x;
  const long n_my = (t0 < a.n_pix_tiles) ? (a.n_pix_tiles - t0 + stride - 1) / stride : 0;
  int s_ty0 = 0, s_tx0 = 0;
  if (n_my > 0) {
    set_tile(t0);
    static_for<N_ITEMS>([&](auto qc) { load_item(qc); });
    static_for<N_ITEMS>([&](auto qc) { store_item(qc, ty0, tx0, smem); });
  }
  if (n_my > 1) {
    set_tile(t0 + stride);
    s_ty0 = ty0;
    s_tx0 = tx0;
    static_for<N_ITEMS>([&](auto qc) { load_item(qc); });
  }
  __syncthreads();
#ifdef UNETPP_WQ_EXP_CLOCK
  const unsigned long long clk0 = __builtin_amdgcn_s_memtime(), ref0 = __builtin_amdgcn_s_memrealtime();
#endif
  // (UNETPP_WQ_EXP_*: timing experiments of tools/wgrad_quad_ablation.sh -- wrong results, never in the shipped build)
  for (long i = 0; i < n_my; ++i) {
    const unsigned cur = static_cast<unsigned>(i & 1) * BUF;
    unsigned char* nbuf = smem + ((i + 1) & 1) * BUF;
    const bool do_store = i + 1 < n_my, do_load = i + 2 < n_my;
    auto hook = [&](auto sc) __attribute__((always_inline)) {
      constexpr int slot = decltype(sc)::v;
      if constexpr (slot < N_ITEMS) {
#ifndef UNETPP_WQ_EXP_NO_STORE
        if (do_store) store_item(IC<slot>{}, s_ty0, s_tx0, nbuf);
#endif
      } else if constexpr (slot < 2 * N_ITEMS) {
#ifndef UNETPP_WQ_EXP_NO_LOAD
        if (do_load) {
          if constexpr (slot == N_ITEMS) {
            set_tile(t0 + (i + 2) * stride);
            s_ty0 = ty0;
            s_tx0 = tx0;
          }
          load_item(IC<slot - N_ITEMS>{});
        }
#endif
      }
    };
#ifdef UNETPP_WQ_EXP_INTERLEAVE
    compute(cur, hook);
#elif defined(UNETPP_WQ_EXP_NO_COMPUTE)
    static_for<2 * N_ITEMS>(hook);
#else
    static_for<2 * N_ITEMS>(hook);
    compute(cur, [](auto) {});
#endif
#ifndef UNETPP_WQ_EXP_NO_BARRIER
    __syncthreads();
#endif
  }
#ifdef UNETPP_WQ_EXP_CLOCK
  if (tid == 0) {
    atomicAdd(&g_wq_clock[0], __builtin_amdgcn_s_memtime() - clk0);
    atomicAdd(&g_wq_clock[1], __builtin_amdgcn_s_memrealtime() - ref0);
  }
#endif
#ifdef UNETPP_WQ_EXP_NO_SLAB
  if (a.n_pix_tiles >= 0) return;
#endif

  const long slab_stride = (static_cast<long>(TAPS) * a.Ktot + 1) * a.Ncols;
  float* slab = d.slabs + blockIdx.x * slab_stride;
  const int k0 = kbase + c0 + 32 * kl, n0 = col_base + nc0 + 32 * nl;
  if constexpr (TAPS == 9) {
    // every wave owns its three taps of the pair's block (MFMA D map: reg r of lane (j, h) = row (r&3) + 8*(r>>2) + 4h)
#pragma unroll
    for (int fr = 0; fr < 3; ++fr) {
      const int t = 3 * fr + sub;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        slab[(static_cast<long>(t) * a.Ktot + k0 + row) * a.Ncols + n0 + j] = acc[fr][r];
      }
    }
  } else {
    // the second wave of every pair hands its sums to the first one through LDS
    float* region = reinterpret_cast<float*>(smem) + pair * 1024 + lane * 4;
    if (sub == 1) {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        *reinterpret_cast<f32x4*>(region + q * 256) = f32x4{acc[0][4 * q], acc[0][4 * q + 1], acc[0][4 * q + 2], acc[0][4 * q + 3]};
    }
    __syncthreads();
    if (sub == 0) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(region + q * 256);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[0][4 * q + i] += v[i];
      }
      store_slab_block<1>(acc, slab, a.Ktot, a.Ncols, k0, 32, n0, 32, j, h);
    }
  }
  if (want_db) {   // uniform: per-thread column sums -> [thread][8] in LDS -> 64 columns, fixed order
    __syncthreads();
    float* red = reinterpret_cast<float*>(smem) + 4 * 1024;   // past the 1x1 pair sums
#pragma unroll
    for (int e = 0; e < 8; ++e) red[tid * 8 + e] = dbacc[e];
    __syncthreads();
    if (tid < 64) {   // column tid = octet (tid >> 3), element (tid & 7): threads octet + 8m
      float sum = 0.f;
      for (int m = 0; m < THREADS / 8; ++m) sum += red[((tid >> 3) + 8 * m) * 8 + (tid & 7)];
      slab[static_cast<long>(TAPS) * a.Ktot * a.Ncols + col_base + nc0 + tid] = sum;
    }
  }
}

template <int TAPS, int LOG2TW>
int launch_quad(const WBfArgs& a, dim3 grid, hipStream_t st) {
  constexpr int XPIX = (TAPS == 9) ? kMaxHaloPixels : kBlockPixels;
  constexpr int BUF = 2 * (XPIX * 64 + 128) + 2 * (kBlockPixels * 64 + 128);
  constexpr size_t lds = 2 * BUF + 512;
  static_assert(lds <= 160 * 1024, "one workgroup per CU");
  static std::atomic<unsigned long long> opted_in[4] = {};
  int device = 0;
  if (hipGetDevice(&device) != hipSuccess || device < 0 || device > 255) return UNETPP_ELAUNCH;
  const unsigned long long bit = 1ull << (device & 63);
  if (!(opted_in[device >> 6].load(std::memory_order_acquire) & bit)) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_bf16_quad_kernel<TAPS, LOG2TW>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess)
      return UNETPP_ELAUNCH;
    opted_in[device >> 6].fetch_or(bit, std::memory_order_release);
  }
  hipLaunchKernelGGL((wgrad_bf16_quad_kernel<TAPS, LOG2TW>), grid, dim3(QuadShape<TAPS>::THREADS), lds, st, a);
  note_kernel(TAPS == 9 ? "wgrad_bf16_quad_kernel<9>" : "wgrad_bf16_quad_kernel<1>");
  return launch_status();
}

}  // namespace

// the quad kernel takes a layer when every view is a multiple of 64 channels wide (UNETPP_BF16_WGRAD_QUAD=0 keeps the
// pair kernel, for A/B runs and the tests that compare the two)
bool wgrad_bf16_quads(const unetpp_wgrad_desc* d) {
  // (unetpp_debug_set("BF16_WGRAD_QUAD", 0): the tests compare the quad and the pair kernel inside one process)
  if (opt_value(OPT_BF16_WGRAD_QUAD, 1) == 0 || !(d->flags & UNETPP_GEMM_BF16)) return false;
  if (d->taps == 9 && d->n_x == 1 && d->x[0].c_len <= 4) return false;  // first layer: its own kernel
  auto fits = [&](const unetpp_view& v) { return v.c_len > 0 && (v.c_len & 63) == 0; };
  for (int i = 0; i < d->n_x; ++i)
    if (!fits(d->x[i])) return false;
  for (int i = 0; i < d->n_dy; ++i)
    if (!fits(d->dy[i])) return false;
  return true;
}

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
  for (int i = 0; i < d->n_x; ++i)   // 31-bit byte offsets inside an image (buffer resources, one per image)
    if (static_cast<long>(d->x[i].Hs) * d->x[i].Ws * d->x[i].C * 2 > 0x7fffffffL) return UNETPP_EINVAL;
  for (int i = 0; i < d->n_dy; ++i)
    if (static_cast<long>(d->dy[i].Hs) * d->dy[i].Ws * d->dy[i].C * 2 > 0x7fffffffL) return UNETPP_EINVAL;
  if (wgrad_bf16_quads(d)) {
    int kq = 0, nq = 0;
    for (int i = 0; i < d->n_x; ++i) kq += d->x[i].c_len >> 6;
    for (int i = 0; i < d->n_dy; ++i) nq += d->dy[i].c_len >> 6;
    a.n_tiles_cols = nq;
    const dim3 qgrid(static_cast<unsigned>(d->n_split), static_cast<unsigned>(kq * nq));
    if (d->taps == 9) {
      if (g.log2tw == 5) return launch_quad<9, 5>(a, qgrid, st);
      if (g.log2tw == 4) return launch_quad<9, 4>(a, qgrid, st);
      return launch_quad<9, 3>(a, qgrid, st);
    }
    if (g.log2tw == 5) return launch_quad<1, 5>(a, qgrid, st);
    if (g.log2tw == 4) return launch_quad<1, 4>(a, qgrid, st);
    return launch_quad<1, 3>(a, qgrid, st);
  }
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

#ifdef UNETPP_WQ_EXP_CLOCK
extern "C" int unetpp_wq_clock_read(unsigned long long* out2, int reset) {
  if (hipMemcpyFromSymbol(out2, HIP_SYMBOL(unetpp::g_wq_clock), 16) != hipSuccess) return 1;
  if (reset) {
    const unsigned long long z[2] = {0, 0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(unetpp::g_wq_clock), z, 16) != hipSuccess) return 1;
  }
  return 0;
}
#endif

// Winograd F(2x2, 3x3) form of the 3x3 weight gradient (same split-K-over-pixel-patches scheme and slab protocol as
// wgrad_dma.hip, 2.25x fewer MFMA cycles):
//
//   dg = G^T [ sum_tiles (B^T d B) (.) (A dY A^T) ] G      d = 4x4 window of x, dY = 2x2 tile of dy
//
// The kernel accumulates the transform-domain products dU[xi][c][n] (16 planes instead of 9 taps);
// unetpp_wgrad_finish sums the slabs and applies G^T . G.  Mapping onto v_mfma_f32_16x16x4_f32:
//   * MFMA row = input channel, MFMA column = output column, MFMA k = 4 TILES (the contraction runs over tiles);
//   * 512 threads = 8 waves = 2 channel halves (16 of the 32 channels of the k-tile) x 4 tile groups; a wave owns
//     4 of the 16 tile columns of the 256-pixel patch (4 MFMA k-steps) and the 16 x 32 block of all 16 dU planes:
//     acc[16][2] float4 = 128 VGPRs;
//   * A operand: lane (channel c, tile slot g) reads its tile's 4x4 window at channel c (16 ds_read_b32), runs
//     B^T d B in registers (32 add/sub) -> 16 values; B operand: lane (column n, tile slot g) reads the 2x2 dy tile and
//     runs A dY A^T without the sign flips of A's last row (12 add/sub; the finish kernel applies the signs);
//   * patches go HBM/L2 -> LDS by global_load_lds (two distinct static buffers, loop unrolled by two, as in
//     wgrad_dma.hip).  LDS image: 8-pixel groups of 1 KB (one DMA instruction each) separated by a small pad; the 4
//     tile slots of an MFMA k-step are the 4 tile ROWS of the patch at one tile column, i.e. pixels a whole number of
//     groups apart (x rows are 36 pixels in LDS), so every window address is a per-lane constant plus an immediate
//     offset and the pads rotate the slots' banks apart (2-way conflicts at worst);
//   * x views may carry the folded BatchNorm apply + ReLU of their producer (applied at the window read; NaN padding);
//   * epilogue: the 4 tile groups are summed in a fixed-order tree through LDS, one slab per workgroup; db comes
//     from the (1,1) element of A dY A^T, which is the plain sum of the 2x2 tile.
#include <cstdlib>

#include "common.h"
#include "lds_asm.h"

namespace unetpp {
namespace {

// In-kernel phase stamps (profiling builds only: -DUNETPP_WWINO_STAMPS, tools/wino_stamps.py --wgrad)
#ifdef UNETPP_WWINO_STAMPS
__device__ unsigned long long g_wwino_stamps[16];
#define WW_STAMP(i)                            \
  do {                                         \
    const unsigned long long now_ = clock64(); \
    st_acc[i] += now_ - st_last;               \
    st_last = now_;                            \
  } while (0)
#else
#define WW_STAMP(i) \
  do {              \
  } while (0)
#endif

constexpr int kWThreads = 512;
constexpr int kTW = 32, kTH = 8, kHWp = kTW + 2, kHHp = kTH + 2;
constexpr int kXRow = 36;                          // LDS row stride of the x patch in pixels (34 + 2 unused): two rows
                                                   // = 72 pixels = 9 whole groups
constexpr int kNPix = kXRow * kHHp;                // 360 pixel slots
constexpr int GX = 264;                            // floats per 8-pixel group of the x patch (256 + 8 pad)
constexpr int GY = 260;                            // ... of the dy patch (256 + 4 pad)
constexpr int XG = kNPix / 8;                      // 45 groups
constexpr int YG = kBlockPixels / 8;               // 32 groups
constexpr int X_FLOATS = XG * GX;                  // 11880
constexpr int BUF = X_FLOATS + YG * GY;            // 20200 floats = 80800 B per buffer
constexpr int X_ITEMS = (XG * 64 + kWThreads - 1) / kWThreads;  // 6
constexpr int DY_ITEMS = YG * 64 / kWThreads;                   // 4

struct WWinoArgs {
  unetpp_wgrad_desc d;
  int tiles_x, tiles_y;
  int Ktot, Ncols, n_tiles_cols;
  long n_pix_tiles;
  int rsrc_ok;  // every view's image is below 2 GB: operands through per-image buffer resources (plain-view kernel)
};

constexpr int x_slot(int U) { return (U >> 3) * GX + (U & 7) * 32; }  // float offset of pixel slot U of the x patch

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// XFORM: every x view carries the folded BatchNorm apply + ReLU (all or none: mixed launches take wgrad_fast.hip)
template <bool XFORM>
__global__ __launch_bounds__(kWThreads, 2) void wgrad_wino_kernel(const WWinoArgs a) {
  __shared__ __attribute__((aligned(16))) float buf_a[BUF];
  __shared__ __attribute__((aligned(16))) float buf_b[BUF];

  const unetpp_wgrad_desc& d = a.d;
  const int tid = threadIdx.x;
  // (wave index through readfirstlane: a scalar, so that the LDS-DMA destinations and the wave-uniform tests below are
  // scalar arithmetic -- as a vector value every DMA piece paid a v_mad + v_readfirstlane for its M0)
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, t16 = lane & 15, g = lane >> 4;
  const int ch = wave & 1;                                         // channel half
  const int tg = wave >> 1;                                        // tile group: tile columns 4*tg .. 4*tg + 3

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

  // ---- staging: item it = (group it>>6, pixel (it>>3)&7 of the group, channel quad it&7); the group is wave uniform
  // per item, the quad and the pixel-in-group are the same for every item of a thread ----
  const int cc = (tid & 7) << 2;
  const bool kx_ok = cc < k_cnt, nx_ok = cc < n_cnt;
  unsigned xdelta[X_ITEMS], ydelta[DY_ITEMS];
  unsigned x_valid = 0;  // bit q: item q is a real slot (columns 34, 35 of a row are not) of a channel quad inside the view
#pragma unroll
  for (int q = 0; q < X_ITEMS; ++q) {
    const int hp = (tid >> 3) + q * (kWThreads >> 3);
    const int hy = hp / kXRow, hx = min(hp - hy * kXRow, kHWp - 1);  // slots 34, 35 of a row are never loaded
    xdelta[q] = static_cast<unsigned>(((hy * X.sy) * X.Ws + hx * X.sx) * X.C + cc) * 4u;
    if (cc < k_cnt && hp - hy * kXRow < kHWp) x_valid |= 1u << q;
  }
#pragma unroll
  for (int q = 0; q < DY_ITEMS; ++q) {
    const int p = (tid >> 3) + q * (kWThreads >> 3);
    ydelta[q] = static_cast<unsigned>((((p >> 5) * DY.sy) * DY.Ws + (p & 31) * DY.sx) * DY.C + cc) * 4u;
  }
  // x views may carry the producer's BatchNorm apply + ReLU (scale/shift + relu): it is applied when the window is
  // read.  Zero padding must stay zero AFTER that transform, so out-of-image pixels are filled with NaN instead:
  // fma(NaN, s, t) = NaN and v_max_f32(NaN, 0) = 0.  Channels past the slice get scale = shift = 0.
  const float x_pad = XFORM ? __builtin_nanf("") : 0.f;
  float x_sc = 0.f, x_sh = 0.f;
  if (XFORM && 16 * ch + t16 < k_cnt) {
    x_sc = X.scale[c0 + 16 * ch + t16];
    x_sh = X.shift[c0 + 16 * ch + t16];
  }
  // Edge patches take two passes: zero fills of out-of-image pixels first (plain ds_writes), then the DMAs -- a
  // ds_write into an array with a DMA in flight makes hipcc drain vmcnt first (see wgrad_dma.hip).
  // in_block: hipcc hoists the zero extension of a 32-bit offset out of the loop and then keeps ten 64-bit pairs (and
  // adds them to the base with v_lshl_add_u64); an offset that is "produced" next to its use stays one register and
  // selects the DMA's scalar-base + 32-bit-offset addressing.
  auto in_block = [](unsigned v) {
    asm volatile("" : "+v"(v));
    return v;
  };
  // interior patches: a slot that is never loaded (columns 34, 35 of a row; channel quads past a narrow view) carries the
  // out-of-range marker in its delta, so issuing an item is the DMA instruction alone: constant vector offset, the
  // patch origin in the scalar offset (round 6: the select + add + produced-in-place copy per item, ten items per patch,
  // and the two divisions of the tile decode were ~80 vector instructions per patch beside MFMAs that do not overlap them)
  unsigned xdelta_i[X_ITEMS], ydelta_i[DY_ITEMS];
#pragma unroll
  for (int q = 0; q < X_ITEMS; ++q) xdelta_i[q] = ((x_valid >> q) & 1u) ? xdelta[q] : 0x80000000u;
#pragma unroll
  for (int q = 0; q < DY_ITEMS; ++q) ydelta_i[q] = nx_ok ? ydelta[q] : 0x80000000u;
  // border patches of images that are whole patches wide and high (W % 32 == 0, H % 8 == 0: every level of the network):
  // an x slot lies outside the image only in the first / last halo column or row of a patch at the matching image
  // edge -- four bits per item (hx == 0, hx == 33, hy == 0, hy == 9), tested against the patch's four edge flags
  unsigned edge_bits = 0;
#pragma unroll
  for (int q = 0; q < X_ITEMS; ++q) {
    const int hp = (tid >> 3) + q * (kWThreads >> 3);
    const int hy = hp / kXRow, hx = hp - hy * kXRow;
    edge_bits |= static_cast<unsigned>((hx == 0) | ((hx == kHWp - 1) << 1) | ((hy == 0) << 2) | ((hy == kHHp - 1) << 3)) << (4 * q);
  }
  const bool whole_patches = (d.W & (kTW - 1)) == 0 && (d.H & (kTH - 1)) == 0;
  // The tiles of a workgroup are blockIdx.x, + gridDim.x, ...: the (image, patch row, patch column) cursor of the next
  // tile to issue is stepped by the decomposed stride on the scalar unit instead of being divided out per tile.
  int c_tx, c_ty, c_n, s_tx, s_ty, s_n;
  {
    unsigned b = blockIdx.x;
    c_tx = static_cast<int>(b % static_cast<unsigned>(a.tiles_x));
    b /= static_cast<unsigned>(a.tiles_x);
    c_ty = static_cast<int>(b % static_cast<unsigned>(a.tiles_y));
    c_n = static_cast<int>(b / static_cast<unsigned>(a.tiles_y));
    b = gridDim.x;
    s_tx = static_cast<int>(b % static_cast<unsigned>(a.tiles_x));
    b /= static_cast<unsigned>(a.tiles_x);
    s_ty = static_cast<int>(b % static_cast<unsigned>(a.tiles_y));
    s_n = static_cast<int>(b / static_cast<unsigned>(a.tiles_y));
    c_tx = __builtin_amdgcn_readfirstlane(c_tx);
    c_ty = __builtin_amdgcn_readfirstlane(c_ty);
    c_n = __builtin_amdgcn_readfirstlane(c_n);
    s_tx = __builtin_amdgcn_readfirstlane(s_tx);
    s_ty = __builtin_amdgcn_readfirstlane(s_ty);
    s_n = __builtin_amdgcn_readfirstlane(s_n);
  }
  auto issue_tile = [&](unsigned, float* buf) {  // the tiles are issued in order: the cursor names the one meant
    __builtin_amdgcn_s_setprio(3);
    const int txi = c_tx, tyi = c_ty, n = c_n;
    {  // step the cursor (all uniform)
      c_tx += s_tx;
      const int cx = c_tx >= a.tiles_x ? 1 : 0;
      c_tx -= cx * a.tiles_x;
      c_ty += s_ty + cx;
      const int cy = c_ty >= a.tiles_y ? 1 : 0;
      c_ty -= cy * a.tiles_y;
      c_n += s_n + cy;
    }
    const int ty0 = tyi * kTH, tx0 = txi * kTW;
    const char* xb = reinterpret_cast<const char*>(X.ptr + view_pixel_offset(X, n, ty0 - 1, tx0 - 1) + c0);
    const char* yb = reinterpret_cast<const char*>(DY.ptr + view_pixel_offset(DY, n, ty0, tx0) + nc0);
#ifdef UNETPP_WWINO_EXP_OLD_STAGING   // A/B builds only (tools/wino_ablation.sh)
    constexpr bool kRsrcStaging = false;
#else
    constexpr bool kRsrcStaging = !XFORM;
#endif
    if constexpr (kRsrcStaging) {
      {
        // Plain views (images below 2 GB: launcher): both operands by LDS-DMA through a buffer resource over the tile's image.
        // An item outside the image (or a channel quad past a narrow view) gets an offset past the end of the image
        // and the DMA writes zeros -- one straight path for interior and border patches.  (The two-pass border path
        // below -- range tests, predicated zero fills, predicated DMAs -- runs for 30 % of the patches of a 256 x 256
        // image, 56 % at 128 x 128 and every patch at 64 x 64.)
        const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(X.ptr) + static_cast<long>(n) * X.Hs * X.Ws * X.C, 0, X.Hs * X.Ws * X.C * 4, 0x00020000);
        const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(
            const_cast<float*>(DY.ptr) + static_cast<long>(n) * DY.Hs * DY.Ws * DY.C, 0, DY.Hs * DY.Ws * DY.C * 4, 0x00020000);
        const unsigned xorg =
            static_cast<unsigned>((((ty0 - 1) * X.sy + X.oy) * X.Ws + (tx0 - 1) * X.sx + X.ox) * X.C + X.c_off + c0) * 4u;
        const unsigned yorg = static_cast<unsigned>(((ty0 * DY.sy + DY.oy) * DY.Ws + tx0 * DY.sx + DY.ox) * DY.C + DY.c_off + nc0) * 4u;
        // interior patches (uniform test) skip the per-item range tests: ~15 vector instructions per item, and with
        // ten items per wave and patch they were 5 % of the kernel (tools/wwino_ablation.sh)
        const bool interior = ty0 >= 1 && tx0 >= 1 && ty0 + kTH + 1 <= d.H && tx0 + kTW + 1 <= d.W;
        if (interior) {
#pragma unroll
          for (int q = 0; q < X_ITEMS; ++q) {
            if (q * 8 + wave < XG)  // wave uniform: the last item ends after 45 groups
              __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lptr_t)(buf + (q * 8 + wave) * GX), 16, static_cast<int>(xdelta_i[q]),
                                                       static_cast<int>(xorg), 0, 0);
          }
#pragma unroll
          for (int q = 0; q < DY_ITEMS; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (lptr_t)(buf + X_FLOATS + (q * 8 + wave) * GY), 16, static_cast<int>(ydelta_i[q]),
                                                     static_cast<int>(yorg), 0, 0);
        } else if (whole_patches) {
          const unsigned em = static_cast<unsigned>((tx0 == 0) | ((tx0 + kTW == d.W) << 1) | ((ty0 == 0) << 2) | ((ty0 + kTH == d.H) << 3)) * 0x111111u;
          const unsigned hit = edge_bits & em;
#pragma unroll
          for (int q = 0; q < X_ITEMS; ++q) {
            if (q * 8 + wave < XG) {
              const unsigned off = ((hit >> (4 * q)) & 15u) ? 0x80000000u : xorg + xdelta_i[q];  // (xorg may be "negative": it wraps back)
              __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lptr_t)(buf + (q * 8 + wave) * GX), 16, static_cast<int>(off), 0, 0, 0);
            }
          }
#pragma unroll
          for (int q = 0; q < DY_ITEMS; ++q)  // (the dy patch has no halo: inside the image)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (lptr_t)(buf + X_FLOATS + (q * 8 + wave) * GY), 16, static_cast<int>(ydelta_i[q]),
                                                     static_cast<int>(yorg), 0, 0);
        } else {
          const int pix = static_cast<int>(in_block(static_cast<unsigned>(tid))) >> 3;  // pixel slot of item 0; item q sits 64 slots on
#pragma unroll
          for (int q = 0; q < X_ITEMS; ++q) {
            if (q * 8 + wave < XG) {
              const int hp = pix + q * (kWThreads >> 3);
              const int hy = hp / kXRow, hx = hp - hy * kXRow;
              const int y = ty0 + hy - 1, x = tx0 + hx - 1;
              const bool inside = kx_ok & (hx < kHWp) & (static_cast<unsigned>(y) < static_cast<unsigned>(d.H)) &
                                  (static_cast<unsigned>(x) < static_cast<unsigned>(d.W));
              const unsigned off = in_block(inside ? xorg + xdelta[q] : 0x80000000u);  // produced next to its use
              __builtin_amdgcn_raw_ptr_buffer_load_lds(xr, (lptr_t)(buf + (q * 8 + wave) * GX), 16, static_cast<int>(off), 0, 0, 0);
            }
          }
#pragma unroll
          for (int q = 0; q < DY_ITEMS; ++q) {
            const int p = pix + q * (kWThreads >> 3);
            const bool inside = nx_ok & (ty0 + (p >> 5) < d.H) & (tx0 + (p & 31) < d.W);
            const unsigned off = in_block(inside ? yorg + ydelta[q] : 0x80000000u);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(yr, (lptr_t)(buf + X_FLOATS + (q * 8 + wave) * GY), 16, static_cast<int>(off), 0, 0, 0);
          }
        }
        __builtin_amdgcn_s_setprio(0);
        return;
      }
    }
    if constexpr (!kRsrcStaging) {
    // views with a folded BatchNorm apply (NaN padding, see above): interior patches straight, border patches in two passes
    const bool interior = ty0 >= 1 && tx0 >= 1 && ty0 + kTH + 1 <= d.H && tx0 + kTW + 1 <= d.W;
    const int lq = (tid & 63) * 4;  // float slot of this lane inside its 1 KB group
    if (interior) {
      if (kx_ok) {
#pragma unroll
        for (int q = 0; q < X_ITEMS; ++q) {
          float* lbase = buf + (q * 8 + wave) * GX;  // wave-uniform; the DMA adds lane * 16 bytes
          const int hp = (tid >> 3) + q * (kWThreads >> 3);
          if (hp < kNPix && hp % kXRow < kHWp)
            __builtin_amdgcn_global_load_lds((gptr_t)(xb + in_block(xdelta[q])), (lptr_t)lbase, 16, 0, 0);
        }
      }
      if (nx_ok) {
#pragma unroll
        for (int q = 0; q < DY_ITEMS; ++q) {
          float* lbase = buf + X_FLOATS + (q * 8 + wave) * GY;
          __builtin_amdgcn_global_load_lds((gptr_t)(yb + in_block(ydelta[q])), (lptr_t)lbase, 16, 0, 0);
        }
      }
    } else {
      unsigned xin = 0, yin = 0;  // bit q: item q of this thread lies inside the image
      // everything below derives from te: "produced" here, so that hipcc cannot hoist two dozen per-thread
      // invariants of this (border-only) path out of the tile loop into registers the MFMA phase needs
      const int te = static_cast<int>(in_block(static_cast<unsigned>(tid)));
      const int we = te >> 6, lqe = (te & 63) * 4;
#pragma unroll
      for (int q = 0; q < X_ITEMS; ++q) {
        const int it = te + q * kWThreads;
        const int hp = it >> 3;
        const int hy = hp / kXRow, hx = hp - hy * kXRow;
        const int y = ty0 + hy - 1, x = tx0 + hx - 1;
        const bool valid = hp < kNPix && hx < kHWp && kx_ok;
        const bool inimg = y >= 0 && y < d.H && x >= 0 && x < d.W;
        if (valid && inimg) xin |= 1u << q;
        if (valid && !inimg) *reinterpret_cast<f32x4*>(buf + (q * 8 + we) * GX + lqe) = f32x4{x_pad, x_pad, x_pad, x_pad};
      }
#pragma unroll
      for (int q = 0; q < DY_ITEMS; ++q) {
        const int p = (te + q * kWThreads) >> 3;
        const bool inimg = ty0 + (p >> 5) < d.H && tx0 + (p & 31) < d.W;
        if (nx_ok && inimg) yin |= 1u << q;
        if (nx_ok && !inimg) *reinterpret_cast<f32x4*>(buf + X_FLOATS + (q * 8 + we) * GY + lqe) = f32x4{0.f, 0.f, 0.f, 0.f};
      }
#pragma unroll
      for (int q = 0; q < X_ITEMS; ++q) {
        float* lbase = buf + (q * 8 + wave) * GX;
        if ((xin >> q) & 1u) __builtin_amdgcn_global_load_lds((gptr_t)(xb + in_block(xdelta[q])), (lptr_t)lbase, 16, 0, 0);
      }
#pragma unroll
      for (int q = 0; q < DY_ITEMS; ++q) {
        float* lbase = buf + X_FLOATS + (q * 8 + wave) * GY;
        if ((yin >> q) & 1u) __builtin_amdgcn_global_load_lds((gptr_t)(yb + in_block(ydelta[q])), (lptr_t)lbase, 16, 0, 0);
      }
    }
    }
    __builtin_amdgcn_s_setprio(0);
  };

  f32x4 acc[16][2];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi)
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) acc[xi][nh] = f32x4{0.f, 0.f, 0.f, 0.f};
  float dbsum[2] = {0.f, 0.f};

  // One staged patch: this wave's k-step r covers tile column 4*tg + r; MFMA k-slot g = tile row g.  Pixel slot of a
  // window element = 72 g + 8 tg + (36 i + j + 2 r): whole groups for g and tg, a compile-time rest.
  const int lane_x = (9 * g + tg) * GX + 16 * ch + t16;
  const int lane_y = X_FLOATS + (8 * g + tg) * GY + t16;
  // Adjacent windows of a tile row share two columns, and the column pass of B^T d B works on whole columns: k-step r
  // reads only the window's last column pair (ds_read2_b32: the two pixels are 128 bytes apart in their group), runs
  // the column pass on it and keeps the result for k-step r + 1 (tc).  All adds are packed (v_pk_add_f32 on the
  // register pairs the reads deliver); every value is the same expression of the same operands as in the 16-read form.
  f32x2_t tc[4];
  auto compute = [&](const float* buf, auto hc) {
    const unsigned xb = lds_offset(buf) + lane_x * 4, yb = lds_offset(buf) + lane_y * 4;
    static_for<2>([&](auto rrc) {
      constexpr int r = 2 * decltype(hc)::v + decltype(rrc)::v;
      // all LDS operands of the k-step by hand-placed reads with ONE wait (lds_asm.h): left to hipcc, each group is
      // read directly in front of its first use.  x: single reads with 16-bit immediates off one address (rows are
      // 4.5 groups apart, out of ds_read2's 8-bit range); dy: the two pixels of a tile row in one ds_read2_b32.
      f32x2_t in[8], first[4];  // in[i]: columns 2r+2, 2r+3 of window row i; in[4 + 2 nh + ap]: dy row ap, column half nh
      static_for<4>([&](auto ic) {
        constexpr int i = decltype(ic)::v, U = kXRow * i + 2 * r + 2;
        float lo, hi;
        lds_read_b32<x_slot(U) * 4>(lo, xb);
        lds_read_b32<x_slot(U + 1) * 4>(hi, xb);
        in[i] = f32x2_t{lo, hi};
        if (r == 0) {
          lds_read_b32<x_slot(kXRow * i) * 4>(lo, xb);
          lds_read_b32<x_slot(kXRow * i + 1) * 4>(hi, xb);
          first[i] = f32x2_t{lo, hi};
        }
      });
      const unsigned yb1 = in_block(yb) + 4 * GY * 4;  // dy row 1: four groups up (kept out of the long-lived registers)
      static_for<4>([&](auto ic) {
        constexpr int nh = decltype(ic)::v >> 1, ap = decltype(ic)::v & 1;
        lds_read2_b32<16 * nh + 2 * r * 32, 16 * nh + (2 * r + 1) * 32>(in[4 + 2 * nh + ap], ap ? yb1 : yb);
      });
      if (r == 0) lds_wait4x2(first);
      lds_wait8x2(in);
      auto fold = [&](f32x2_t& v) {  // BatchNorm apply + ReLU of the producer, folded into the operand read;
        v.x = fmaxf(fmaf(v.x, x_sc, x_sh), 0.f);  // max(NaN, 0) = 0 (padding)
        v.y = fmaxf(fmaf(v.y, x_sc, x_sh), 0.f);
      };
      auto column_pass = [&](f32x2_t* x, f32x2_t* t) {
        if (XFORM) {
#pragma unroll
          for (int i = 0; i < 4; ++i) fold(x[i]);
        }
        t[0] = pk_sub(x[0], x[2]);
        t[1] = pk_add(x[1], x[2]);
        t[2] = pk_sub(x[2], x[1]);
        t[3] = pk_sub(x[1], x[3]);
      };
      if (r == 0) column_pass(first, tc);
      f32x2_t tn[4], V[8];
      column_pass(in, tn);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        V[2 * i] = pk_sub_add_x(tc[i], tn[i]);      // t0 - t2, t1 + t2
        V[2 * i + 1] = pk_cross_sub(tc[i], tn[i]);  // t2 - t1, t1 - t3
        tc[i] = tn[i];
      }
      // The packed adds are inline asm, which hipcc's hazard recogniser does not treat as VALU writes: one of them
      // placed among or right behind the MFMAs may overwrite a register that an MFMA in flight still reads as its C
      // operand (the accumulators are renamed freely; 7 wait states are required).  All of them depend on the k-step's
      // LDS reads (a round trip behind the previous MFMA batch); the second column half's additionally on an s_nop 7
      // that "produces" their inputs, and fences keep them out of the MFMA batches.
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        // A dY A^T with A's last row taken as (0, +1): rows (d0, d0 + d1, d0 - d1, d1); the finish kernel applies the
        // sign (-1)^[a == 3] (-1)^[b == 3]
        if (nh == 1) asm volatile("s_nop 7" : "+v"(in[6]), "+v"(in[7]));
        const f32x2_t rw[4] = {in[4 + 2 * nh], pk_add(in[4 + 2 * nh], in[5 + 2 * nh]),
                               pk_sub(in[4 + 2 * nh], in[5 + 2 * nh]), in[5 + 2 * nh]};
        f32x2_t sd[4];  // (left + right, left - right) of each row
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) sd[aa] = pk_sum_diff(rw[aa]);
        dbsum[nh] += sd[1].x;  // (1,1): the plain sum of the 2x2 tile
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int aa = 0; aa < 4; ++aa) {
          const float M[4] = {rw[aa].x, sd[aa].x, sd[aa].y, rw[aa].y};
#pragma unroll
          for (int bb = 0; bb < 4; ++bb) {
            const int xi = 4 * aa + bb;
            const float v = (xi & 1) ? V[xi >> 1].y : V[xi >> 1].x;
            acc[xi][nh] = __builtin_amdgcn_mfma_f32_16x16x4f32(v, M[bb], acc[xi][nh], 0, 0, 0);
          }
        }
        __builtin_amdgcn_sched_barrier(0);  // 128 accumulator registers: keep the next batch's operands from being hoisted
      }
    });
  };

  // tiles of this workgroup: blockIdx.x, +gridDim.x, ...   (buffer A holds even, buffer B odd local tiles)
  const unsigned stride = gridDim.x;
  const unsigned t0 = blockIdx.x;
  const unsigned n_tiles_all = static_cast<unsigned>(a.n_pix_tiles);
  const int n_my = (t0 < n_tiles_all) ? static_cast<int>((n_tiles_all - t0 + stride - 1) / stride) : 0;
#ifdef UNETPP_WWINO_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = clock64();
#endif
  if (n_my > 0) issue_tile(t0, buf_a);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  WW_STAMP(0);  // 0: prologue
  // Waves w and w+4 share a SIMD: waves 0-3 issue the next patch's DMAs before their MFMAs, waves 4-7 half way through.
  const bool late = wave >= 4;
  for (int i = 0; i < n_my; i += 2) {
    if (!late && i + 1 < n_my) issue_tile(t0 + (i + 1) * stride, buf_b);
    WW_STAMP(1);  // 1: DMA issue (early waves)
    compute(buf_a, IC<0>{});
    WW_STAMP(2);  // 2: MFMA half 0
    if (late && i + 1 < n_my) issue_tile(t0 + (i + 1) * stride, buf_b);
    WW_STAMP(3);  // 3: DMA issue (late waves)
    compute(buf_a, IC<1>{});
    WW_STAMP(4);  // 4: MFMA half 1
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    WW_STAMP(5);  // 5: wait for the DMAs
    __syncthreads();
    WW_STAMP(6);  // 6: barrier
    if (i + 1 < n_my) {
      if (!late && i + 2 < n_my) issue_tile(t0 + (i + 2) * stride, buf_a);
      WW_STAMP(1);
      compute(buf_b, IC<0>{});
      WW_STAMP(2);
      if (late && i + 2 < n_my) issue_tile(t0 + (i + 2) * stride, buf_a);
      WW_STAMP(3);
      compute(buf_b, IC<1>{});
      WW_STAMP(4);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      WW_STAMP(5);
      __syncthreads();
      WW_STAMP(6);
    }
  }

#ifdef UNETPP_WWINO_STAMPS
  if (lane == 0) {
    for (int i = 0; i < 8; ++i) atomicAdd(&g_wwino_stamps[i], st_acc[i]);
    atomicAdd(&g_wwino_stamps[8], 1ull);
  }
#endif
  // ---- fixed-order tree over the 4 tile groups of each channel half: (tg0 + tg2) + (tg1 + tg3).  A region holds
  // the 32 float4 accumulators of a wave lane-linearly (32 KB); two regions per buffer. ----
  constexpr int R = 32 * 64 * 4;  // floats per region
  auto region = [&](float* base) { return base + ch * R + lane * 4; };
  auto put = [&](float* rg) {
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) *reinterpret_cast<f32x4*>(rg + (xi * 2 + nh) * 256) = acc[xi][nh];
  };
  auto add = [&](const float* rg) {
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
      for (int nh = 0; nh < 2; ++nh) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(rg + (xi * 2 + nh) * 256);
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[xi][nh][e] += v[e];
        if (nh == 1 && (xi & 3) == 3) __builtin_amdgcn_sched_barrier(0);  // at most 8 loads in flight
      }
  };
  if (tg == 2) put(region(buf_a));
  if (tg == 3) put(region(buf_b));
  __syncthreads();
  if (tg == 0) add(region(buf_a));
  if (tg == 1) add(region(buf_b));
  __syncthreads();
  if (tg == 1) put(region(buf_a));
  __syncthreads();
  if (tg == 0) add(region(buf_a));

  // ---- one slab per workgroup: [K][Ncols][16 planes] (the 16 planes of a (channel, column) pair are one 64-byte
  // run: 16-byte stores here, coalesced reads in the finish kernel), then the db row ----
  const long slab_stride = (16L * a.Ktot + 1) * a.Ncols;
  float* slab = d.slabs + blockIdx.x * slab_stride;
  if (tg == 0) {
    // register e of acc[xi][nh] of lane (t16, g): channel 16*ch + 4*g + e, column 16*nh + t16
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      const int n = 16 * nh + t16;
      if (n < n_cnt) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int c = 16 * ch + 4 * g + e;
          if (c < k_cnt) {
            float* dst = slab + (static_cast<long>(kbase + c0 + c) * a.Ncols + n0 + n) * 16;
#pragma unroll
            for (int q = 0; q < 4; ++q)
              *reinterpret_cast<f32x4*>(dst + 4 * q) =
                  f32x4{acc[4 * q][nh][e], acc[4 * q + 1][nh][e], acc[4 * q + 2][nh][e], acc[4 * q + 3][nh][e]};
          }
        }
      }
    }
  }
  if (want_db) {
    __syncthreads();  // the regions are dead
    float* dbs = buf_b;  // [tg 4][32 columns]
#pragma unroll
    for (int nh = 0; nh < 2; ++nh) {
      dbsum[nh] += __shfl_xor(dbsum[nh], 16);
      dbsum[nh] += __shfl_xor(dbsum[nh], 32);
      if (ch == 0 && g == 0) dbs[tg * 32 + 16 * nh + t16] = dbsum[nh];
    }
    __syncthreads();
    if (tid < n_cnt)
      slab[16L * a.Ktot * a.Ncols + n0 + tid] = (dbs[tid] + dbs[64 + tid]) + (dbs[32 + tid] + dbs[96 + tid]);
  }
}

// slab sum + G^T . G: 1024 threads = SG slab groups x EPB consecutive slab floats (EPB / 16 (k, n) pairs x 16 planes);
// SG = 2^log2_sg <= 16 by the number of slabs (see wgrad_finish_kernel); the bias row is summed by extra blocks
__global__ void wgrad_finish_wino_kernel(const float* __restrict__ slabs, int n_split, int K, int Ncols,
                                         float* __restrict__ dw, long d_t, long d_k, long d_n, float* __restrict__ db,
                                         int log2_sg) {
  const int SG = 1 << log2_sg, EPB = 1024 >> log2_sg;  // group g sums slabs g, g + SG, ... -- all loads independent
  __shared__ float part[1024];                          // [SG][EPB]
  const long pairs = static_cast<long>(K) * Ncols;
  const long total = (16L * K + 1) * Ncols;
  const int ppb = EPB >> 4;  // (k, n) pairs per block
  const long pair_blocks = (pairs + ppb - 1) / ppb;
  const int e = threadIdx.x & (EPB - 1), sg = threadIdx.x >> (10 - log2_sg);
  if (static_cast<long>(blockIdx.x) >= pair_blocks) {  // bias row: 64 columns per block
    const long n = (blockIdx.x - pair_blocks) * 64L + e;
    float s = 0.f;
    if (e < 64 && n < Ncols)
      for (int b = sg; b < n_split; b += SG) s += slabs[static_cast<long>(b) * total + 16L * pairs + n];
    part[sg * EPB + e] = s;
    __syncthreads();
    if (sg == 0 && e < 64 && n < Ncols && db != nullptr) {
      float t = 0.f;
      for (int q = 0; q < SG; ++q) t += part[q * EPB + e];
      db[n] = t;
    }
    return;
  }
  const long pair = blockIdx.x * static_cast<long>(ppb) + (e >> 4);
  float s = 0.f;
  if (pair < pairs) {
    const long i = pair * 16 + (e & 15);  // EPB consecutive floats per block
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int b = sg;
    for (; b + 3 * SG < n_split; b += 4 * SG) {
      s0 += slabs[static_cast<long>(b) * total + i];
      s1 += slabs[static_cast<long>(b + SG) * total + i];
      s2 += slabs[static_cast<long>(b + 2 * SG) * total + i];
      s3 += slabs[static_cast<long>(b + 3 * SG) * total + i];
    }
    for (; b < n_split; b += SG) s0 += slabs[static_cast<long>(b) * total + i];
    s = (s0 + s1) + (s2 + s3);
  }
  part[sg * EPB + e] = s;
  __syncthreads();
  if (sg == 0) {
    float t = 0.f;
    for (int q = 0; q < SG; ++q) t += part[q * EPB + e];  // fixed order
    part[e] = t;
  }
  __syncthreads();
  if (static_cast<int>(threadIdx.x) >= 9 * ppb || dw == nullptr) return;
  const int pl = threadIdx.x % ppb, tap = threadIdx.x / ppb;  // 9 taps x ppb pairs
  const long p = blockIdx.x * static_cast<long>(ppb) + pl;
  if (p >= pairs) return;
  const int r = tap / 3, c = tap - 3 * r;
  // G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]]; column r of G, with the sign of the kernel's unsigned last row
  const float gr[4] = {r == 0 ? 1.f : 0.f, 0.5f, (r == 1) ? -0.5f : 0.5f, r == 2 ? -1.f : 0.f};
  const float gc[4] = {c == 0 ? 1.f : 0.f, 0.5f, (c == 1) ? -0.5f : 0.5f, c == 2 ? -1.f : 0.f};
  float out = 0.f;
#pragma unroll
  for (int aa = 0; aa < 4; ++aa) {
    float rowsum = 0.f;
#pragma unroll
    for (int bb = 0; bb < 4; ++bb) rowsum = fmaf(gc[bb], part[pl * 16 + aa * 4 + bb], rowsum);
    out = fmaf(gr[aa], rowsum, out);
  }
  const long k = p / Ncols, n = p - k * Ncols;
  dw[tap * d_t + k * d_k + n * d_n] = out;
}

bool plain_aligned(const unetpp_view& v) {
  return v.scale == nullptr && v.gate == nullptr && !v.relu && ((v.C | v.c_off | v.c_len) & 3) == 0 &&
         (reinterpret_cast<uintptr_t>(v.ptr) & 15) == 0;
}

// x views: plain, or with the folded BatchNorm load transform (scale + shift + relu together)
bool x_view_ok(const unetpp_view& v) {
  if (v.gate != nullptr || ((v.C | v.c_off | v.c_len) & 3) != 0 || (reinterpret_cast<uintptr_t>(v.ptr) & 15) != 0) return false;
  return v.scale == nullptr ? !v.relu : (v.shift != nullptr && v.relu != 0);
}

}  // namespace

#ifdef UNETPP_WWINO_STAMPS
extern "C" int unetpp_debug_wwino_stamps(unsigned long long* out16, int reset) {  // profiling builds only
  if (out16 != nullptr && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_wwino_stamps), sizeof(g_wwino_stamps)) != hipSuccess)
    return UNETPP_ELAUNCH;
  if (reset) {
    const unsigned long long zero[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wwino_stamps), zero, sizeof(zero)) != hipSuccess) return UNETPP_ELAUNCH;
  }
  return UNETPP_OK;
}
#endif

// 3x3, Winograd not forbidden, 32-wide patches, every view plain and 16-byte aligned, 32-bit byte offsets inside a patch
bool wgrad_wino_applies(const unetpp_wgrad_desc* d) {
  if (d == nullptr || d->taps != 9 || (d->flags & UNETPP_GEMM_DIRECT) != 0) return false;
  if (d->N <= 0 || d->H <= 0 || d->W <= 0 || tile_geom(d->H, d->W).log2tw != 5) return false;
  if (d->n_x < 1 || d->n_x > UNETPP_MAX_VIEWS || d->n_dy < 1 || d->n_dy > UNETPP_MAX_VIEWS) return false;
  for (int i = 0; i < d->n_x; ++i) {
    const unetpp_view& v = d->x[i];
    if (!view_ok(v) || !x_view_ok(v) || v.c_len < 8) return false;
    if ((v.scale != nullptr) != (d->x[0].scale != nullptr)) return false;  // all views transformed, or none
    if (static_cast<long>(kHHp) * v.sy * v.Ws * v.C * 4 >= 0x7fffffffL) return false;
  }
  for (int i = 0; i < d->n_dy; ++i) {
    const unetpp_view& v = d->dy[i];
    if (!view_ok(v) || !plain_aligned(v)) return false;
    if (static_cast<long>(kHHp) * v.sy * v.Ws * v.C * 4 >= 0x7fffffffL) return false;
  }
  return true;
}

// returns UNETPP_OK after launching, or 1 when the descriptor needs another kernel
int launch_wgrad_wino(const unetpp_wgrad_desc* d, int Ktot, int Ncols, int n_tiles_cols, int k_tiles, hipStream_t st) {
  if (!wgrad_wino_applies(d)) return 1;
  WWinoArgs a;
  a.d = *d;
  a.Ktot = Ktot;
  a.Ncols = Ncols;
  a.n_tiles_cols = n_tiles_cols;
  const TileGeom g = tile_geom(d->H, d->W);
  a.tiles_x = g.tiles_x;
  a.tiles_y = g.tiles_y;
  a.n_pix_tiles = static_cast<long>(d->N) * g.tiles_y * g.tiles_x;
  if (a.n_pix_tiles >= 0x7fffffffL) return 1;  // 32-bit tile indices in the kernel
  a.rsrc_ok = 1;
  for (int i = 0; i < d->n_x; ++i)
    if (static_cast<long>(d->x[i].Hs) * d->x[i].Ws * d->x[i].C * 4 > 0x7fffffffL) a.rsrc_ok = 0;
  for (int i = 0; i < d->n_dy; ++i)
    if (static_cast<long>(d->dy[i].Hs) * d->dy[i].Ws * d->dy[i].C * 4 > 0x7fffffffL) a.rsrc_ok = 0;
  const bool xform = d->x[0].scale != nullptr;  // all views or none (wgrad_wino_applies)
#ifndef UNETPP_WWINO_EXP_OLD_STAGING
  if (!xform && !a.rsrc_ok) return 1;  // plain views above 2 GB per image: the direct-sum kernels
#endif
  const dim3 grid(static_cast<unsigned>(d->n_split), static_cast<unsigned>(static_cast<long>(k_tiles) * n_tiles_cols));
  if (d->x[0].scale != nullptr)
    hipLaunchKernelGGL(wgrad_wino_kernel<true>, grid, dim3(kWThreads), 0, st, a);
  else
    hipLaunchKernelGGL(wgrad_wino_kernel<false>, grid, dim3(kWThreads), 0, st, a);
  note_kernel("wgrad_wino_kernel");
  return launch_status();
}

int launch_wgrad_finish_wino(const float* slabs, int n_split, int K, int Ncols, float* dw, long d_t, long d_k, long d_n,
                             float* db, hipStream_t st) {
  const long pairs = static_cast<long>(K) * Ncols;
  const int log2_sg = finish_log2_groups(n_split);
  const long ppb = (1024 >> log2_sg) >> 4;  // (k, n) pairs per block
  const unsigned blocks = static_cast<unsigned>((pairs + ppb - 1) / ppb + (Ncols + 63) / 64);
  hipLaunchKernelGGL(wgrad_finish_wino_kernel, dim3(blocks), dim3(1024), 0, st, slabs, n_split, K, Ncols, dw, d_t, d_k,
                     d_n, db, log2_sg);
  return launch_status();
}

}  // namespace unetpp

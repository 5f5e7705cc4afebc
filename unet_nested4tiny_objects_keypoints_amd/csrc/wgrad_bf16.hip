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

// In-kernel phase stamps (profiling builds only: -DUNETPP_WBF_STAMPS, tools/wbf_stamps.py): wave 0 of every workgroup adds
// the s_memtime cycles it spent in each phase of its tile loop to a global table.
#ifdef UNETPP_WBF_STAMPS
__device__ unsigned long long g_wbf_stamps[16];
#define WBF_STAMP(i)                           \
  do {                                         \
    const unsigned long long now_ = clock64(); \
    st_acc[i] += now_ - st_last;               \
    st_last = now_;                            \
  } while (0)
#else
#define WBF_STAMP(i) \
  do {               \
  } while (0)
#endif

template <int OFF>
__device__ __forceinline__ void lds_read_tr(u32x2& v, unsigned addr) {
  asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF) : "memory");
}
__device__ __forceinline__ void lds_wait2(u32x2& a, u32x2& b) { asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a), "+v"(b)); }

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// DMA = the views carry no load transform: the tiles go HBM/L2 -> LDS by LDS-DMA (global_load_lds, 16 bytes per lane,
// lane-linear -- exactly the [pixel][32 channels] rows the transposed reads want) into a ring of THREE buffers, two
// tiles ahead of the MFMAs, without staging registers: a tile is 18 MFMAs (0.25 us) per wave against ~2 us of memory
// latency, so the register path (one tile ahead, kept for x views with the folded BatchNorm transform) is latency bound.
template <int TAPS, int LOG2TW, bool DMA>
__global__ __launch_bounds__(kWThreads, 1) void wgrad_bf16_kernel(const WBfArgs a) {
  constexpr int HALO = (TAPS == 9) ? 1 : 0;
  constexpr int TW = 1 << LOG2TW, TH = kBlockPixels >> LOG2TW;
  constexpr int HWp = TW + 2 * HALO, HHp = TH + 2 * HALO;
  constexpr int NPIX = HWp * HHp;
  constexpr int XPIX = (TAPS == 9) ? kMaxHaloPixels : kBlockPixels;
  constexpr int X_ITEMS = (NPIX * 4 + kWThreads - 1) / kWThreads;  // 16-byte items (8 channels), 4 per pixel: <= 3
  constexpr int DY_ITEMS = (kBlockPixels * 4) / kWThreads;         // 2
  constexpr int N_ITEMS = X_ITEMS + DY_ITEMS;
  // the DMA path issues every item of every lane (uniform instruction counts for s_waitcnt): room for the padding items
  constexpr int X_BYTES = DMA ? X_ITEMS * kWThreads * 16 : XPIX * 64;
  constexpr int DY_BYTES = kBlockPixels * 64;
  constexpr int BUF = X_BYTES + DY_BYTES;
  constexpr int NBUF = DMA ? 3 : 2;
  constexpr int TREE_BYTES = 4 * TAPS * 4096;                      // four regions of TAPS*1024 floats
  constexpr int TILE_BYTES = (NBUF * BUF > TREE_BYTES) ? NBUF * BUF : TREE_BYTES;
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
  if (!DMA && (k_cnt < 32 || n_cnt < 32)) {  // (the DMA path loads clamped channels there: rows / columns nobody stores)
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
  auto set_tile = [&](long tile) {  // tile < 2^31 (launcher): 32-bit divisions (a 64-bit one costs ~300 cycles, three of
    unsigned b = static_cast<unsigned>(tile);  // them per tile were most of a tile's 2 us)
    const unsigned txi = b % static_cast<unsigned>(a.tiles_x);
    b /= static_cast<unsigned>(a.tiles_x);
    const unsigned tyi = b % static_cast<unsigned>(a.tiles_y);
    img = static_cast<int>(b / static_cast<unsigned>(a.tiles_y));
    ty0 = static_cast<int>(tyi) * TH;
    tx0 = static_cast<int>(txi) * TW;
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
  if constexpr (DMA) {
    // every lane issues all N_ITEMS DMAs of a tile from a valid (clamped) address; out-of-image pixels are zeroed by
    // plain LDS stores once the tile has landed (edge tiles only)
    // Interior tiles (all but the image border): an item's address is the tile origin (wave uniform, scalar registers)
    // plus a per-thread offset that never changes -- one 64-bit add per DMA instead of ~50 instructions of clamping
    // and index arithmetic.  A tile's MFMAs take 0.25 us per wave: instruction issue, not memory, bounded the loop.
    long rel[N_ITEMS];  // element offset of the item from the tile's first (halo) pixel, incl. its channel octet
#pragma unroll
    for (int qi = 0; qi < N_ITEMS; ++qi) {
      const bool isx = qi < X_ITEMS;
      const int it = tid + (isx ? qi : qi - X_ITEMS) * kWThreads;
      const int cc = (it & 3) << 3;
      if (isx) {
        const int hp = min(it >> 2, NPIX - 1);
        const int hy = hp / HWp, hx = hp - hy * HWp;
        rel[qi] = (static_cast<long>(hy) * X.sy * X.Ws + static_cast<long>(hx) * X.sx) * X.C + (cc < k_cnt ? cc : 0);
      } else {
        const int p = it >> 2;
        rel[qi] = (static_cast<long>(p >> LOG2TW) * DY.sy * DY.Ws + static_cast<long>(p & (TW - 1)) * DY.sx) * DY.C +
                  (cc < n_cnt ? cc : 0);
      }
    }
    auto issue_tile = [&](long tile, unsigned char* buf) {
      set_tile(tile);
      const bool interior = ty0 >= HALO && tx0 >= HALO && ty0 + TH + HALO <= d.H && tx0 + TW + HALO <= d.W;
      if (interior) {  // uniform
        const bf16_t* xo = xptr + view_pixel_offset(X, img, ty0 - HALO, tx0 - HALO) + c0;
        const bf16_t* yo = dyptr + view_pixel_offset(DY, img, ty0, tx0) + nc0;
        auto uniform_ptr = [](const bf16_t* p) {  // both halves through readfirstlane (it returns int: no sign extension)
          const uintptr_t u = reinterpret_cast<uintptr_t>(p);
          const unsigned lo = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<unsigned>(u))));
          const unsigned hi = static_cast<unsigned>(__builtin_amdgcn_readfirstlane(static_cast<int>(static_cast<unsigned>(u >> 32))));
          return (static_cast<uintptr_t>(hi) << 32) | static_cast<uintptr_t>(lo);
        };
        const uintptr_t xo_u = uniform_ptr(xo), yo_u = uniform_ptr(yo);
#pragma unroll
        for (int qi = 0; qi < N_ITEMS; ++qi) {
          const bool isx = qi < X_ITEMS;
          const bf16_t* src = reinterpret_cast<const bf16_t*>(isx ? xo_u : yo_u) + rel[qi];
          unsigned char* lbase = buf + (isx ? 0 : X_BYTES) + ((isx ? qi : qi - X_ITEMS) * kWThreads + wave * 64) * 16;
          __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lbase, 16, 0, 0);
        }
        return;
      }
#pragma unroll
      for (int qi = 0; qi < N_ITEMS; ++qi) {
        const bool isx = qi < X_ITEMS;
        const int it = tid + (isx ? qi : qi - X_ITEMS) * kWThreads;
        const bf16_t* src;
        if (isx) {
          const int hp = min(it >> 2, NPIX - 1), cc = (it & 3) << 3;
          const int hy = hp / HWp, hx = hp - hy * HWp;
          const int y = min(max(ty0 + hy - HALO, 0), d.H - 1), x = min(max(tx0 + hx - HALO, 0), d.W - 1);
          src = xptr + view_pixel_offset(X, img, y, x) + c0 + (cc < k_cnt ? cc : 0);
        } else {
          const int p = it >> 2, cc = (it & 3) << 3;
          const int y = min(ty0 + (p >> LOG2TW), d.H - 1), x = min(tx0 + (p & (TW - 1)), d.W - 1);
          src = dyptr + view_pixel_offset(DY, img, y, x) + nc0 + (cc < n_cnt ? cc : 0);
        }
        unsigned char* lbase = buf + (isx ? 0 : X_BYTES) + ((isx ? qi : qi - X_ITEMS) * kWThreads + wave * 64) * 16;
        __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)lbase, 16, 0, 0);
      }
    };
    auto fix_edges = [&](long tile, unsigned char* buf) {
      set_tile(tile);
      const bool interior = ty0 >= HALO && tx0 >= HALO && ty0 + TH + HALO <= d.H && tx0 + TW + HALO <= d.W;
      if (interior) return;  // uniform
#pragma unroll
      for (int qi = 0; qi < X_ITEMS; ++qi) {
        const int it = tid + qi * kWThreads;
        const int hp = it >> 2;
        const int hy = hp / HWp, hx = hp - hy * HWp;
        const int y = ty0 + hy - HALO, x = tx0 + hx - HALO;
        if (it < NPIX * 4 && !(y >= 0 && y < d.H && x >= 0 && x < d.W)) *reinterpret_cast<u32x4*>(&buf[it * 16]) = u32x4{0u, 0u, 0u, 0u};
      }
#pragma unroll
      for (int qi = 0; qi < DY_ITEMS; ++qi) {
        const int it = tid + qi * kWThreads;
        const int p = it >> 2;
        if (!(ty0 + (p >> LOG2TW) < d.H && tx0 + (p & (TW - 1)) < d.W)) *reinterpret_cast<u32x4*>(&buf[X_BYTES + it * 16]) = u32x4{0u, 0u, 0u, 0u};
      }
    };
#ifdef UNETPP_WBF_STAMPS
    unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    unsigned long long st_last = clock64();
#endif
    if (n_my > 0) issue_tile(t0, smem);
    if (n_my > 1) issue_tile(t0 + stride, smem + BUF);
    const int n_mine = static_cast<int>(n_my);
    WBF_STAMP(0);  // 0: prologue
    for (int i = 0; i < n_mine; ++i) {
      unsigned char* cur = smem + (i % 3) * BUF;
      // tile i has landed once at most the DMAs of tile i+1 are outstanding (vmcnt counts this wave's DMAs in order)
      if (i + 1 < n_mine) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_ITEMS) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      WBF_STAMP(1);  // 1: wait for the tile's DMAs
      fix_edges(t0 + i * stride, cur);
      WBF_STAMP(2);  // 2: edge fix-up
      // every wave's share of tile i is in LDS; every wave is done with tile i-1 (buffer (i+2) % 3).  A raw s_barrier:
      // __syncthreads() carries a fence, and hipcc drains vmcnt(0) for it -- LDS-DMA writes count there -- which would
      // also wait for tile i+1 and collapse the ring to depth one.
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      WBF_STAMP(3);  // 3: barrier
#ifndef UNETPP_WBF_EXP_NO_DMA  // experiment builds (tools/README.md)
      if (i + 2 < n_mine) issue_tile(t0 + (i + 2) * stride, smem + ((i + 2) % 3) * BUF);
#endif
      WBF_STAMP(4);  // 4: DMA issue
#ifndef UNETPP_WBF_EXP_NO_COMPUTE
      compute(static_cast<unsigned>(i % 3) * BUF);
#endif
      WBF_STAMP(5);  // 5: transposed reads + MFMAs
    }
    __syncthreads();
#ifdef UNETPP_WBF_STAMPS
    if (tid == 0) {
      for (int i = 0; i < 8; ++i) atomicAdd(&g_wbf_stamps[i], st_acc[i]);
      atomicAdd(&g_wbf_stamps[8], static_cast<unsigned long long>(n_mine));
      atomicAdd(&g_wbf_stamps[9], 1ull);
    }
#endif
  } else {
    // Register path, prefetch distance two with ONE register stage and two LDS buffers: at the top of iteration i the
    // registers hold tile i+1 (requested a whole iteration ago); they go into the buffer tile i-1 was read from, tile
    // i+2 is requested, then tile i's MFMAs run.  One barrier per tile.
    u32x4 stage[N_ITEMS];
    int s_ty0 = 0, s_tx0 = 0;  // geometry of the tile held in `stage` (store_item reads ty0 / tx0)
    if (n_my > 0) {
      set_tile(t0);
#pragma unroll
      for (int qi = 0; qi < N_ITEMS; ++qi) store_item(qi, smem, load_item(qi));
    }
    if (n_my > 1) {
      set_tile(t0 + stride);
      s_ty0 = ty0;
      s_tx0 = tx0;
#pragma unroll
      for (int qi = 0; qi < N_ITEMS; ++qi) stage[qi] = load_item(qi);
    }
    __syncthreads();
    for (long i = 0; i < n_my; ++i) {
      const unsigned cur = static_cast<unsigned>(i & 1) * BUF;
      if (i + 1 < n_my) {
        ty0 = s_ty0;
        tx0 = s_tx0;
        unsigned char* nxt = smem + ((i + 1) & 1) * BUF;  // last read by tile i-1, a barrier ago
#pragma unroll
        for (int qi = 0; qi < N_ITEMS; ++qi) store_item(qi, nxt, stage[qi]);
      }
      if (i + 2 < n_my) {
        set_tile(t0 + (i + 2) * stride);
        s_ty0 = ty0;
        s_tx0 = tx0;
#pragma unroll
        for (int qi = 0; qi < N_ITEMS; ++qi) stage[qi] = load_item(qi);
      }
      compute(cur);
      __syncthreads();
    }
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

template <int TAPS, int LOG2TW, bool DMA>
int launch_one(const WBfArgs& a, dim3 grid, hipStream_t st) {
  constexpr int HALO = (TAPS == 9) ? 1 : 0;
  constexpr int TW = 1 << LOG2TW, TH = kBlockPixels >> LOG2TW;
  constexpr int NPIX = (TW + 2 * HALO) * (TH + 2 * HALO);
  constexpr int XPIX = (TAPS == 9) ? kMaxHaloPixels : kBlockPixels;
  constexpr int X_ITEMS = (NPIX * 4 + kWThreads - 1) / kWThreads;
  constexpr int BUF = (DMA ? X_ITEMS * kWThreads * 16 : XPIX * 64) + kBlockPixels * 64;
  constexpr int NBUF = DMA ? 3 : 2;
  constexpr int TREE = 4 * TAPS * 4096;
  constexpr size_t lds = ((NBUF * BUF > TREE) ? NBUF * BUF : TREE) + 256;
  static_assert(lds <= 160 * 1024, "LDS budget");
  // > 64 KB of dynamic LDS needs the per-function opt-in: once per process and kernel (the call costs tens of
  // microseconds of host time, more than a short launch runs on the device)
  static bool opted_in = false;
  if (!opted_in) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(wgrad_bf16_kernel<TAPS, LOG2TW, DMA>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess)
      return UNETPP_ELAUNCH;
    opted_in = true;
  }
  hipLaunchKernelGGL((wgrad_bf16_kernel<TAPS, LOG2TW, DMA>), grid, dim3(kWThreads), lds, st, a);
  note_kernel(TAPS == 9 ? "wgrad_bf16_kernel<9>" : "wgrad_bf16_kernel<1>");
  return launch_status();
}

}  // namespace

#ifdef UNETPP_WBF_STAMPS
extern "C" int unetpp_debug_wbf_stamps(unsigned long long* out16, int reset) {  // profiling builds only
  if (out16 != nullptr && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_wbf_stamps), sizeof(g_wbf_stamps)) != hipSuccess)
    return UNETPP_ELAUNCH;
  if (reset) {
    const unsigned long long zero[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wbf_stamps), zero, sizeof(zero)) != hipSuccess) return UNETPP_ELAUNCH;
  }
  return UNETPP_OK;
}
#endif

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
  const dim3 grid(static_cast<unsigned>(d->n_split), static_cast<unsigned>(static_cast<long>(k_tiles) * n_tiles_cols));
  bool dma = true;  // plain views only
  for (int i = 0; i < d->n_x; ++i) dma = dma && d->x[i].scale == nullptr && !d->x[i].relu;
#ifdef UNETPP_WBF_EXP_REGISTER_PATH  // experiment builds: the register path for every launch
  dma = false;
#endif
#define UNETPP_WBF(T, L)                                                   \
  return dma ? launch_one<T, L, true>(a, grid, st) : launch_one<T, L, false>(a, grid, st)
  if (d->taps == 9) {
    if (g.log2tw == 5) UNETPP_WBF(9, 5);
    if (g.log2tw == 4) UNETPP_WBF(9, 4);
    UNETPP_WBF(9, 3);
  }
  if (g.log2tw == 5) UNETPP_WBF(1, 5);
  if (g.log2tw == 4) UNETPP_WBF(1, 4);
  UNETPP_WBF(1, 3);
#undef UNETPP_WBF
}

}  // namespace unetpp

// The network's first convolution (models/unet.py:220, conv00.conv1): 1..4 input channels -> f0 output channels.
// K = 9*Cin <= 36 is far too short for the MFMA tile (the generic kernel pads it to 72 and spends 8x the work), and
// the layer is HBM-bound anyway: forward writes 32 channels per pixel from 1-3 read, the weight gradient reads them.
// Plain VALU kernels, one 256-pixel patch at a time (the same patch geometry as the MFMA kernels, so the BatchNorm
// partial rows line up):
//   forward : input patch + halo and the 9*Cin*Cout weights in LDS; thread = (pixel, 4 output channels), so a wave
//             stores whole 128-byte pixel rows; fused bias, ReLU, BatchNorm partial sums
//   wgrad   : thread = (pixel stripe, 4 output channels) keeps 9*Cin float4 accumulators in registers over a
//             grid-stride loop of patches; fixed-order LDS reduction; one slab per workgroup (wgrad_finish sums them)
#include "bf16_common.h"
#include "bn_fused.h"
#include "common.h"
#include "lds_asm.h"

namespace unetpp {
namespace {

constexpr int kMaxCout = 128;

struct SmallArgs {
  const float* x;      // [N, H, W, CIN]
  const float* w;      // packed [9][CIN][COUT]
  const float* bias;   // [COUT] or null
  float* y;            // forward: out tensor base (+ c_off), pixel stride yC
  const float* dy;     // wgrad: dy tensor base (+ c_off), pixel stride yC
  float* stats;        // forward: [patches][COUT][2] or null
  float* slabs;        // wgrad: [n_split][9*CIN + 1][COUT]
  int N, H, W, COUT, yC, relu;
  int log2tw, tiles_x, tiles_y;
  long n_patches;
  int bn_in_kernel;    // forward: the BatchNorm partial rows are per workgroup (bn_fused.h)
};

template <int CIN>
__device__ __forceinline__ void stage_patch(const SmallArgs& a, float* xs, int n, int ty0, int tx0, int TW, int TH) {
  const int HWp = TW + 2, npix = HWp * (TH + 2);
  for (int it = threadIdx.x; it < npix * CIN; it += kThreads) {
    const int hp = it / CIN, ci = it - hp * CIN;
    const int hy = hp / HWp, hx = hp - hy * HWp;
    const int y = ty0 + hy - 1, x = tx0 + hx - 1;
    float v = 0.f;
    if (y >= 0 && y < a.H && x >= 0 && x < a.W) v = a.x[((static_cast<long>(n) * a.H + y) * a.W + x) * CIN + ci];
    xs[it] = v;
  }
}

// BF: the output (forward) / the output gradient (wgrad) is bf16 storage (UNETPP_GEMM_BF16); the input stays fp32
//
// Forward: persistent workgroups over the patches (grid-stride).  The geometry of a thread's halo items is the same
// for every patch, so it is decoded once; the next patch's input is requested into registers before the current one
// is computed (two LDS buffers, one barrier per patch for the input); a thread keeps its four output channels' weights
// in registers when there is one input channel; the BatchNorm partial sums are reduced with lane shuffles inside a
// wave and in fixed order across the four waves.
template <int CIN, bool BF = false>
__global__ __launch_bounds__(kThreads, CIN >= 3 ? 3 : 4) void small_cin_fwd_kernel(const SmallArgs a) {
  constexpr int ITEMS = (kMaxHaloPixels * CIN + kThreads - 1) / kThreads;
  // weights of this thread's four output channels in registers: 9 * CIN float4.  (Round 4: also for 2 and 3 input channels
  // -- from LDS the 27 16-byte weight reads per pixel and thread of the 3-channel layer kept the LDS array busier than the
  // VALU: 110 us for the 82 MB of configs[4]'s first layer.)
  constexpr bool WREG = CIN <= 3;
  __shared__ float xs2[2][kMaxHaloPixels * CIN];
  __shared__ __attribute__((aligned(16))) float ws[WREG ? 4 : 9 * CIN * kMaxCout];
  __shared__ float red[4][kMaxCout * 2];  // [wave][quad][s1 x 4, s2 x 4]
  const int tid = threadIdx.x;
  const bool bn_fused = a.bn_in_kernel != 0;  // uniform
  float run1 = 0.f, run2 = 0.f;               // this thread's column (tid < COUT) over all patches of the workgroup
  const int TW = 1 << a.log2tw, TH = kBlockPixels >> a.log2tw, HWp = TW + 2;
  const int npix = HWp * (TH + 2);
  const int QN = a.COUT >> 2;             // 4-channel groups; the launcher guarantees 256 % QN == 0
  const int quad = tid % QN, psub = tid / QN, pstep = kThreads / QN;

  // halo items of this thread: offset from the patch's first pixel and position inside the halo
  int rel[ITEMS], pos[ITEMS];  // pos = hy << 16 | hx, or -1 for no item
#pragma unroll
  for (int q = 0; q < ITEMS; ++q) {
    const int it = tid + q * kThreads;
    const int hp = it / CIN, ci = it - hp * CIN;
    const int hy = hp / HWp, hx = hp - hy * HWp;
    rel[q] = ((hy - 1) * a.W + (hx - 1)) * CIN + ci;
    pos[q] = hp < npix ? (hy << 16 | hx) : -1;
  }
  float pre[ITEMS];
  auto request = [&](unsigned patch) {
    unsigned b = patch;
    const int txi = static_cast<int>(b % static_cast<unsigned>(a.tiles_x));
    b /= static_cast<unsigned>(a.tiles_x);
    const int tyi = static_cast<int>(b % static_cast<unsigned>(a.tiles_y));
    const int n = static_cast<int>(b / static_cast<unsigned>(a.tiles_y));
    const int ty0 = tyi * TH, tx0 = txi * TW;
    const float* origin = a.x + ((static_cast<long>(n) * a.H + ty0) * a.W + tx0) * CIN;
#pragma unroll
    for (int q = 0; q < ITEMS; ++q) {
      const int y = ty0 + (pos[q] >> 16) - 1, x = tx0 + (pos[q] & 0xffff) - 1;
      pre[q] = (pos[q] >= 0 && y >= 0 && y < a.H && x >= 0 && x < a.W) ? origin[rel[q]] : 0.f;
    }
  };

  f32x4 wr[WREG ? 9 * CIN : 1];
  if (WREG) {
#pragma unroll
    for (int t = 0; t < 9 * CIN; ++t) wr[WREG ? t : 0] = *reinterpret_cast<const f32x4*>(a.w + t * a.COUT + 4 * quad);
  } else {
    for (int i = tid; i < 9 * CIN * a.COUT; i += kThreads) ws[i] = a.w[i];
  }
  f32x4 bias4 = {0.f, 0.f, 0.f, 0.f};
  if (a.bias != nullptr) bias4 = *reinterpret_cast<const f32x4*>(a.bias + 4 * quad);

  const unsigned n_patches = static_cast<unsigned>(a.n_patches);
  unsigned patch = blockIdx.x;
  if (patch < n_patches) request(patch);
  for (int k = 0; patch < n_patches; patch += gridDim.x, ++k) {
    float* xs = xs2[k & 1];
#pragma unroll
    for (int q = 0; q < ITEMS; ++q)
      if (pos[q] >= 0) xs[tid + q * kThreads] = pre[q];
    __syncthreads();  // the patch is staged; everybody is done with the other buffer and with red[]
    if (patch + gridDim.x < n_patches) request(patch + gridDim.x);

    unsigned b = patch;
    const int txi = static_cast<int>(b % static_cast<unsigned>(a.tiles_x));
    b /= static_cast<unsigned>(a.tiles_x);
    const int tyi = static_cast<int>(b % static_cast<unsigned>(a.tiles_y));
    const int n = static_cast<int>(b / static_cast<unsigned>(a.tiles_y));
    const int ty0 = tyi * TH, tx0 = txi * TW;
    float* y_patch = BF ? reinterpret_cast<float*>(reinterpret_cast<bf16_t*>(a.y) + ((static_cast<long>(n) * a.H + ty0) * a.W + tx0) * a.yC)
                        : a.y + ((static_cast<long>(n) * a.H + ty0) * a.W + tx0) * a.yC;
    // two-channel register pairs: hipcc selects v_pk_fma_f32 / v_pk_add_f32 for the vector arithmetic (18 + 4 issue
    // slots per pixel instead of 36 + 8; the kernel is as much VALU- as HBM-bound)
    f32x2_t s1a = {0.f, 0.f}, s1b = {0.f, 0.f}, s2a = {0.f, 0.f}, s2b = {0.f, 0.f};
    for (int p = psub; p < kBlockPixels; p += pstep) {
      const int py = p >> a.log2tw, px = p & (TW - 1);
      const int y = ty0 + py, x = tx0 + px;
      if (y < a.H && x < a.W) {
        f32x2_t acc_a = {bias4[0], bias4[1]}, acc_b = {bias4[2], bias4[3]};
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const float* xp = &xs[((py + tap / 3) * HWp + px + tap % 3) * CIN];
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci) {
            const f32x2_t xv = {xp[ci], xp[ci]};
            const f32x4 w4 = WREG ? wr[WREG ? tap * CIN + ci : 0]
                                  : *reinterpret_cast<const f32x4*>(&ws[WREG ? 0 : (tap * CIN + ci) * a.COUT + 4 * quad]);
            acc_a = __builtin_elementwise_fma(xv, f32x2_t{w4[0], w4[1]}, acc_a);
            acc_b = __builtin_elementwise_fma(xv, f32x2_t{w4[2], w4[3]}, acc_b);
          }
        }
        f32x4 acc = {acc_a.x, acc_a.y, acc_b.x, acc_b.y};
        if (a.relu) {
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = fmaxf(acc[e], 0.f);
        }
        if (BF) {  // the statistics describe the stored (rounded) tensor
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[e] = bf_round(acc[e]);
        }
        const f32x2_t va = {acc[0], acc[1]}, vb = {acc[2], acc[3]};
        s1a += va;
        s1b += vb;
        s2a = __builtin_elementwise_fma(va, va, s2a);
        s2b = __builtin_elementwise_fma(vb, vb, s2b);
        // scalar patch origin + 32-bit offset inside the patch (8 rows of the image at most: the launcher checks)
        const unsigned o = static_cast<unsigned>((py * a.W + px) * a.yC + 4 * quad);
#ifndef UNETPP_FIRST_NO_NT  // streaming stores: the tensor is far larger than the L2, and lines left dirty there delay the next launch (X_0,0 block 428 -> 417 us)
        if (BF)
          __builtin_nontemporal_store(u32x2{pack_bf2(acc[0], acc[1]), pack_bf2(acc[2], acc[3])}, reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(y_patch) + o));
        else
          __builtin_nontemporal_store(acc, reinterpret_cast<f32x4*>(y_patch + o));
#else
        if (BF)
          *reinterpret_cast<u32x2*>(reinterpret_cast<bf16_t*>(y_patch) + o) = u32x2{pack_bf2(acc[0], acc[1]), pack_bf2(acc[2], acc[3])};
        else
          *reinterpret_cast<f32x4*>(y_patch + o) = acc;
#endif
      }
    }
    f32x4 s1 = {s1a.x, s1a.y, s1b.x, s1b.y}, s2 = {s2a.x, s2a.y, s2b.x, s2b.y};
    if (a.stats != nullptr) {  // uniform
      // lanes quad, quad + QN, ... of a wave hold the same channels: butterfly over them, then the four waves in order
      for (int off = QN; off < 64; off <<= 1) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          s1[e] += __shfl_xor(s1[e], off);
          s2[e] += __shfl_xor(s2[e], off);
        }
      }
      const int lane = tid & 63, wave = tid >> 6;
      if (lane < QN) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          red[wave][lane * 8 + e] = s1[e];
          red[wave][lane * 8 + 4 + e] = s2[e];
        }
      }
      __syncthreads();
      if (tid < a.COUT) {
        const int i1 = (tid >> 2) * 8 + (tid & 3);
        const float t1 = (red[0][i1] + red[1][i1]) + (red[2][i1] + red[3][i1]);
        const float t2 = (red[0][i1 + 4] + red[1][i1 + 4]) + (red[2][i1 + 4] + red[3][i1 + 4]);
        if (bn_fused) {
          run1 += t1;
          run2 += t2;
        } else {
          float* dst = a.stats + (static_cast<long>(patch) * a.COUT + tid) * 2;
          dst[0] = t1;
          dst[1] = t2;
        }
      }
    }
  }
  if (bn_fused) {
    __syncthreads();  // everybody is done with red[]: it becomes the workgroup's row
    float* run = &red[0][0];
    if (tid < a.COUT) {
      run[2 * tid] = run1;
      run[2 * tid + 1] = run2;
    }
    __syncthreads();
    bn_rows_store<kThreads>(a.stats, a.COUT, run);
  }
}

template <int CIN, bool BF = false>
__global__ __launch_bounds__(kThreads) void small_cin_wgrad_kernel(const SmallArgs a) {
  __shared__ float xs[kMaxHaloPixels * CIN];
  __shared__ float red[kThreads][4];
  const int tid = threadIdx.x;
  const int TW = 1 << a.log2tw, TH = kBlockPixels >> a.log2tw, HWp = TW + 2;
  const int QN = a.COUT >> 2;
  const int quad = tid % QN, psub = tid / QN, pstep = kThreads / QN;
  f32x4 acc[9 * CIN];
#pragma unroll
  for (int i = 0; i < 9 * CIN; ++i) acc[i] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 db = {0.f, 0.f, 0.f, 0.f};
  for (long patch = blockIdx.x; patch < a.n_patches; patch += gridDim.x) {
    long b = patch;
    const int txi = static_cast<int>(b % a.tiles_x);
    b /= a.tiles_x;
    const int tyi = static_cast<int>(b % a.tiles_y);
    const int n = static_cast<int>(b / a.tiles_y);
    const int ty0 = tyi * TH, tx0 = txi * TW;
    __syncthreads();
    stage_patch<CIN>(a, xs, n, ty0, tx0, TW, TH);
    __syncthreads();
    // Round 5: the dy quads of this thread's pixels are requested in batches of eight BEFORE the arithmetic of the batch.
    // The one-pixel-at-a-time loop exposed a memory round trip per pixel (a dependent load at the head of 36..108 FMAs,
    // 8..32 pixels per thread and patch): the kernel ran at 0.8-1.0 TB/s of its 16 us HBM floor's 5 (configs[4]: 99 us).
    // Same pixels in the same order per thread; a pixel outside the image contributes g = 0 (x * 0 added: no change).
    constexpr int G = (CIN == 4) ? 4 : 8;   // (36 accumulator quads at CIN = 4: a batch of eight does not fit the register file)
    for (int p0 = psub; p0 < kBlockPixels; p0 += G * pstep) {
      f32x4 gq[G];
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int p = p0 + u * pstep;
        const int py = p >> a.log2tw, px = p & (TW - 1);
        const int y = ty0 + py, x = tx0 + px;
        const bool ok = (p < kBlockPixels) & (y < a.H) & (x < a.W);
        // dead items read the patch's first pixel (always inside the image): straight-line loads, no branch
        const long o = ((static_cast<long>(n) * a.H + (ok ? y : ty0)) * a.W + (ok ? x : tx0)) * a.yC + 4 * quad;
        f32x4 g;
        if (BF) {
          const u32x2 v = *reinterpret_cast<const u32x2*>(reinterpret_cast<const bf16_t*>(a.dy) + o);
          g = f32x4{bf_lo(v[0]), bf_hi(v[0]), bf_lo(v[1]), bf_hi(v[1])};
        } else {
          g = *reinterpret_cast<const f32x4*>(a.dy + o);
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) gq[u][e] = ok ? g[e] : 0.f;
      }
#pragma unroll
      for (int u = 0; u < G; ++u) {
        const int p = min(p0 + u * pstep, kBlockPixels - 1);   // (dead items: any staged pixel, times zero)
        const int py = p >> a.log2tw, px = p & (TW - 1);
        const f32x4 g = gq[u];
#pragma unroll
        for (int e = 0; e < 4; ++e) db[e] += g[e];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
          const float* xp = &xs[((py + tap / 3) * HWp + px + tap % 3) * CIN];
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci) {
            const float xv = xp[ci];
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[tap * CIN + ci][e] = fmaf(xv, g[e], acc[tap * CIN + ci][e]);
          }
        }
      }
    }
  }
  // fixed-order reduction over the threads that share an output-channel group, one slab row at a time
  float* slab = a.slabs + static_cast<long>(blockIdx.x) * (9 * CIN + 1) * a.COUT;
#pragma unroll
  for (int row = 0; row <= 9 * CIN; ++row) {
    const f32x4 v = (row < 9 * CIN) ? acc[row < 9 * CIN ? row : 0] : db;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 4; ++e) red[tid][e] = v[e];
    __syncthreads();
    for (int c = tid; c < a.COUT; c += kThreads) {
      const int q = c >> 2, e = c & 3;
      float t = 0.f;
      for (int k = q; k < kThreads; k += QN) t += red[k][e];
      slab[row * a.COUT + c] = t;
    }
  }
}

// Input gradient of the first convolution with bf16 activation storage: dx[n, y, x, ci] = sum_{r, s, co}
// dy[n, y + 1 - r, x + 1 - s, co] * w[co, ci, r, s], dy bf16 NHWC [.., COUT], dx fp32 NHWC [.., CIN <= 4].  A rare path
// (saliency / adversarial inputs: the training step needs no input gradient), so a plain VALU kernel: thread = pixel,
// the 9 x COUT x CIN weights in LDS as [tap][co][ci], dy read in 16-byte pieces straight from L2.
template <int CIN>
__global__ __launch_bounds__(kThreads) void small_cin_dgrad_bf16_kernel(const bf16_t* __restrict__ dy,
                                                                        const float* __restrict__ w, int N, int H, int W,
                                                                        int COUT, float* __restrict__ dx) {
  __shared__ float ws[9 * kMaxCout * CIN];
  for (int i = threadIdx.x; i < 9 * COUT * CIN; i += kThreads) {
    const int ci = i % CIN, co = (i / CIN) % COUT, tap = i / (CIN * COUT);
    ws[i] = w[(static_cast<long>(co) * CIN + ci) * 9 + tap];  // torch layout [co][ci][3][3]
  }
  __syncthreads();
  const long pixels = static_cast<long>(N) * H * W;
  for (long p = blockIdx.x * static_cast<long>(kThreads) + threadIdx.x; p < pixels; p += static_cast<long>(gridDim.x) * kThreads) {
    const int x = static_cast<int>(p % W);
    const long r = p / W;
    const int y = static_cast<int>(r % H);
    const long n = r / H;
    float acc[CIN];
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) acc[ci] = 0.f;
    for (int tap = 0; tap < 9; ++tap) {
      const int yy = y + 1 - tap / 3, xx = x + 1 - tap % 3;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      const u32x4* src = reinterpret_cast<const u32x4*>(dy + ((n * H + yy) * W + xx) * COUT);
      const float* wt = &ws[tap * COUT * CIN];
      for (int c8 = 0; c8 < COUT; c8 += 8) {
        float g[8];
        unpack8(src[c8 >> 3], g);
#pragma unroll
        for (int e = 0; e < 8; ++e)
#pragma unroll
          for (int ci = 0; ci < CIN; ++ci) acc[ci] = fmaf(g[e], wt[(c8 + e) * CIN + ci], acc[ci]);
      }
    }
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) dx[p * CIN + ci] = acc[ci];
  }
}

bool plain_view(const unetpp_view& v) {
  return v.scale == nullptr && v.gate == nullptr && !v.relu && v.sy == 1 && v.sx == 1 && v.oy == 0 && v.ox == 0;
}

bool small_shape_ok(const unetpp_view& x, const unetpp_view& y, int H, int W) {
  if (!plain_view(x) || x.C > 4 || x.c_off != 0 || x.c_len != x.C || x.Hs != H || x.Ws != W) return false;
  if (y.sy != 1 || y.sx != 1 || y.oy != 0 || y.ox != 0 || y.Hs != H || y.Ws != W) return false;
  const int co = y.c_len;
  if ((co & 3) || co > kMaxCout || (kThreads % (co >> 2)) != 0) return false;
  if (((y.C | y.c_off) & 3) || (reinterpret_cast<uintptr_t>(y.ptr) & 15)) return false;
  return true;
}

void fill_geom(SmallArgs& a, int N, int H, int W) {
  const TileGeom g = tile_geom(H, W);
  a.N = N;
  a.H = H;
  a.W = W;
  a.log2tw = g.log2tw;
  a.tiles_x = g.tiles_x;
  a.tiles_y = g.tiles_y;
  a.n_patches = static_cast<long>(N) * g.tiles_y * g.tiles_x;
}

}  // namespace

// 3x3 forward for <= 4 input channels.  Returns 1 when the descriptor does not fit this path.
int launch_small_cin_fwd(const unetpp_gemm_desc* d, hipStream_t st, long* bn_rows) {
  if (d->taps != 9 || d->n_in != 1 || d->n_out != 1) return 1;
  const unetpp_view& X = d->in[0];
  const unetpp_view& Y = d->out[0];
  if (!small_shape_ok(X, Y, d->H, d->W) || Y.gate != nullptr || Y.accumulate) return 1;
  if (d->bias != nullptr && (reinterpret_cast<uintptr_t>(d->bias) & 15)) return 1;
  SmallArgs a = {};
  fill_geom(a, d->N, d->H, d->W);
  if (a.n_patches > 0x7fffffffL) return 1;
  if ((16L * d->W + 64) * Y.C * 4 > 0x7fffffffL) return 1;  // 32-bit offsets inside a patch (at most 16 rows)
  const bool bf = (d->flags & UNETPP_GEMM_BF16) != 0;
  if (d->weight == nullptr) return 1;
  a.x = X.ptr;
  a.w = d->weight;
  a.bias = d->bias;
  a.y = bf ? reinterpret_cast<float*>(reinterpret_cast<bf16_t*>(Y.ptr) + Y.c_off) : Y.ptr + Y.c_off;
  a.stats = d->stats_partial;
  a.COUT = Y.c_len;
  a.yC = Y.C;
  a.relu = Y.relu;
  const int cus = device_cu_count();
  if (cus <= 0) return UNETPP_ELAUNCH;
  // persistent grid: as many workgroups as are resident at once (<= 128 registers up to three input channels, 156
  // with four: four / three one-wave-per-SIMD workgroups per CU)
  long workers = static_cast<long>(cus) * (X.C >= 3 ? 3 : 4);   // = the kernel's launch bounds
  if (workers > kBnFusedRows) workers = kBnFusedRows;
  const dim3 grid(static_cast<unsigned>(a.n_patches < workers ? a.n_patches : workers)), block(kThreads);
  a.bn_in_kernel = bn_rows_per_workgroup(d, a.COUT) ? 1 : 0;
#define UNETPP_SMALL_FWD(B)                                                                          \
  switch (X.C) {                                                                                     \
    case 1: hipLaunchKernelGGL((small_cin_fwd_kernel<1, B>), grid, block, 0, st, a); break;          \
    case 2: hipLaunchKernelGGL((small_cin_fwd_kernel<2, B>), grid, block, 0, st, a); break;          \
    case 3: hipLaunchKernelGGL((small_cin_fwd_kernel<3, B>), grid, block, 0, st, a); break;          \
    default: hipLaunchKernelGGL((small_cin_fwd_kernel<4, B>), grid, block, 0, st, a); break;         \
  }
  if (bf) {
    UNETPP_SMALL_FWD(true)
  } else {
    UNETPP_SMALL_FWD(false)
  }
#undef UNETPP_SMALL_FWD
  note_kernel("small_cin_fwd_kernel");
  if (a.bn_in_kernel && bn_rows != nullptr) *bn_rows = grid.x;
  return launch_status();
}

// weight gradient of the same layer.  Returns 1 when the descriptor does not fit this path.
int launch_small_cin_wgrad(const unetpp_wgrad_desc* d, hipStream_t st) {
  if (d->taps != 9 || d->n_x != 1 || d->n_dy != 1) return 1;
  const unetpp_view& X = d->x[0];
  const unetpp_view& DY = d->dy[0];
  if (!small_shape_ok(X, DY, d->H, d->W) || DY.gate != nullptr || DY.scale != nullptr || DY.relu) return 1;
  SmallArgs a = {};
  fill_geom(a, d->N, d->H, d->W);
  if (d->n_split < 1 || d->n_split > a.n_patches) return 1;
  const bool bf = (d->flags & UNETPP_GEMM_BF16) != 0;
  a.x = X.ptr;
  a.dy = bf ? reinterpret_cast<const float*>(reinterpret_cast<const bf16_t*>(DY.ptr) + DY.c_off) : DY.ptr + DY.c_off;
  a.slabs = d->slabs;
  a.COUT = DY.c_len;
  a.yC = DY.C;
  const dim3 grid(static_cast<unsigned>(d->n_split)), block(kThreads);
#define UNETPP_SMALL_WGRAD(B)                                                                        \
  switch (X.C) {                                                                                     \
    case 1: hipLaunchKernelGGL((small_cin_wgrad_kernel<1, B>), grid, block, 0, st, a); break;        \
    case 2: hipLaunchKernelGGL((small_cin_wgrad_kernel<2, B>), grid, block, 0, st, a); break;        \
    case 3: hipLaunchKernelGGL((small_cin_wgrad_kernel<3, B>), grid, block, 0, st, a); break;        \
    default: hipLaunchKernelGGL((small_cin_wgrad_kernel<4, B>), grid, block, 0, st, a); break;       \
  }
  if (bf) {
    UNETPP_SMALL_WGRAD(true)
  } else {
    UNETPP_SMALL_WGRAD(false)
  }
#undef UNETPP_SMALL_WGRAD
  note_kernel("small_cin_wgrad_kernel");
  return launch_status();
}

}  // namespace unetpp

// dy bf16 [N, H, W, COUT] (COUT a multiple of 8, at most 128), weight fp32 in torch layout [COUT][CIN][3][3], dx fp32 [N, H, W, CIN]
extern "C" int unetpp_first_layer_dgrad_bf16(const void* dy, const float* weight, int32_t N, int32_t H, int32_t W,
                                             int32_t CIN, int32_t COUT, float* dx, void* stream) {
  using namespace unetpp;
  if (!dy || !weight || !dx || N < 1 || H < 1 || W < 1 || CIN < 1 || CIN > 4 || COUT < 8 || (COUT & 7) || COUT > kMaxCout)
    return UNETPP_EINVAL;
  if (reinterpret_cast<uintptr_t>(dy) & 15) return UNETPP_EINVAL;
  const long pixels = static_cast<long>(N) * H * W;
  long blocks = (pixels + kThreads - 1) / kThreads;
  if (blocks > 4096) blocks = 4096;
  const dim3 grid(static_cast<unsigned>(blocks)), block(kThreads);
  hipStream_t st = static_cast<hipStream_t>(stream);
  const bf16_t* g = static_cast<const bf16_t*>(dy);
  switch (CIN) {
    case 1: hipLaunchKernelGGL(small_cin_dgrad_bf16_kernel<1>, grid, block, 0, st, g, weight, N, H, W, COUT, dx); break;
    case 2: hipLaunchKernelGGL(small_cin_dgrad_bf16_kernel<2>, grid, block, 0, st, g, weight, N, H, W, COUT, dx); break;
    case 3: hipLaunchKernelGGL(small_cin_dgrad_bf16_kernel<3>, grid, block, 0, st, g, weight, N, H, W, COUT, dx); break;
    default: hipLaunchKernelGGL(small_cin_dgrad_bf16_kernel<4>, grid, block, 0, st, g, weight, N, H, W, COUT, dx); break;
  }
  return launch_status();
}

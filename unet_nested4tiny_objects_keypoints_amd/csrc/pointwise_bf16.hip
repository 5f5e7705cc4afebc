// HBM-bound companions of the bf16 MFMA kernels (BASELINE configs[3]/[4]): every activation tensor is bf16 NHWC in
// HBM and is touched in 16-byte pieces = 8 channels of one pixel; the arithmetic, the BatchNorm sums and all
// parameter gradients are fp32.
//   unetpp_affine_relu_pool_bf16   BatchNorm apply + ReLU (+ 2x2 max-pool with argmax bytes) of a conv output
//   unetpp_bn_bwd_reduce_bf16      BatchNorm + ReLU backward, pass 1: per-workgroup (sum g, sum g*xhat) partials
//   unetpp_bn_bwd_apply_bf16       pass 2: dy = gamma*invstd*(g - dbeta/M - xhat*dgamma/M)
//                                  (both passes can route the gradient of the node's max-pooled copy to the window
//                                  argmax while they read d_act: no scatter pass, no read-modify-write)
//   unetpp_head_fwd_bf16 / unetpp_head_bwd_bf16   dropout + 1x1 convolution + sigmoid heads: bf16 features in,
//                                  fp32 NCHW probabilities out (the loss stays fp32), bf16 feature gradient back
//   unetpp_maxpool_bwd_bf16        max-pool gradient routed to the window argmax and added to d_act, optionally followed
//                                  by the ReLU mask of the node (is_batchnorm=False: no BatchNorm backward to route through)
//   unetpp_bilinear2x_fwd_bf16 / unetpp_bilinear2x_bwd_bf16   the is_deconv=False up path (align_corners=True), fp32
//                                  interpolation of bf16 values; backward in gather form, optional accumulate + ReLU mask
#include "bf16_common.h"
#include "common.h"
#include "lds_asm.h"
#include "dropout.h"

namespace unetpp {
namespace {

#define ST(s) static_cast<hipStream_t>(s)

inline unsigned grid_for8(long items, long cap = 2048 * 8) {
  long b = (items + kThreads - 1) / kThreads;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return static_cast<unsigned>(b);
}
inline bool a16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

__device__ __forceinline__ void affine8(float (&f)[8], const float* scale, const float* shift, int c, int relu) {
  if (scale != nullptr) {
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], scale[c + e], shift[c + e]);
  }
  if (relu) {
#pragma unroll
    for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
  }
}

// thread = (pixel, channel octet)
__global__ __launch_bounds__(kThreads) void affine_relu_bf16_kernel(const bf16_t* __restrict__ y, const float* scale,
                                                                    const float* shift, int relu, long items, int CG,
                                                                    bf16_t* __restrict__ act) {
  for (long i = blockIdx.x * static_cast<long>(kThreads) + threadIdx.x; i < items;
       i += static_cast<long>(gridDim.x) * kThreads) {
    const int c = static_cast<int>(i % CG) * 8;
    float f[8];
    unpack8(reinterpret_cast<const u32x4*>(y)[i], f);
    affine8(f, scale, shift, c, relu);
    reinterpret_cast<u32x4*>(act)[i] = pack8(f);
  }
}

// thread = (2x2 window, channel octet): four 16-byte loads, four optional stores of the activation, one store of the
// pooled octet and 8 argmax bytes.  The winner is the first maximum in scan order of the ROUNDED (stored) values.
__global__ __launch_bounds__(kThreads) void affine_relu_pool_bf16_kernel(const bf16_t* __restrict__ y, const float* scale,
                                                                         const float* shift, int relu, int N, int H, int W,
                                                                         int CG, bf16_t* __restrict__ act,
                                                                         bf16_t* __restrict__ pooled,
                                                                         uint8_t* __restrict__ pool_idx) {
  const int Hp = H >> 1, Wp = W >> 1;
  const long items = static_cast<long>(N) * Hp * Wp * CG;
  for (long i = blockIdx.x * static_cast<long>(kThreads) + threadIdx.x; i < items;
       i += static_cast<long>(gridDim.x) * kThreads) {
    const unsigned iu = static_cast<unsigned>(i);  // items < 2^31 (launcher)
    const int cg = static_cast<int>(iu % static_cast<unsigned>(CG));
    unsigned r = iu / static_cast<unsigned>(CG);
    const int xp = static_cast<int>(r % static_cast<unsigned>(Wp));
    r /= static_cast<unsigned>(Wp);
    const int yp = static_cast<int>(r % static_cast<unsigned>(Hp));
    const long n = r / static_cast<unsigned>(Hp);
    const long base = ((n * H + 2 * yp) * W + 2 * xp) * CG + cg;  // in octets
    const long offs[4] = {base, base + CG, base + static_cast<long>(W) * CG, base + static_cast<long>(W) * CG + CG};
    float best[8];
    unsigned bi[8];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      float f[8];
      unpack8(reinterpret_cast<const u32x4*>(y)[offs[q]], f);
      affine8(f, scale, shift, cg * 8, relu);
      const u32x4 packed = pack8(f);
      if (act != nullptr) reinterpret_cast<u32x4*>(act)[offs[q]] = packed;
      unpack8(packed, f);  // compare what is stored
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        if (q == 0 || f[e] > best[e]) {
          best[e] = f[e];
          bi[e] = q;
        }
      }
    }
    reinterpret_cast<u32x4*>(pooled)[i] = pack8(best);
    reinterpret_cast<u32x2*>(pool_idx)[i] = u32x2{bi[0] | (bi[1] << 8) | (bi[2] << 16) | (bi[3] << 24),
                                                  bi[4] | (bi[5] << 8) | (bi[6] << 16) | (bi[7] << 24)};
  }
}

// gradient of the activation at (pixel i, octet): d_act (+ the pooled gradient when this pixel won its window)
__device__ __forceinline__ void load_grad8(const bf16_t* d_act, const bf16_t* d_pooled, const uint8_t* pool_idx, long i,
                                           int CG, int H, int W, float (&g)[8]) {
  unpack8(reinterpret_cast<const u32x4*>(d_act)[i], g);
  if (d_pooled != nullptr) {  // 32-bit index arithmetic: items < 2^31 (launchers); CG is a power of two
    const unsigned iu = static_cast<unsigned>(i), ucg = static_cast<unsigned>(CG);
    const unsigned cg = iu & (ucg - 1u);
    unsigned r = iu / ucg;
    const unsigned x = r % static_cast<unsigned>(W);
    r /= static_cast<unsigned>(W);
    const unsigned y = r % static_cast<unsigned>(H);
    const unsigned n = r / static_cast<unsigned>(H);
    const unsigned wi = ((n * (static_cast<unsigned>(H) >> 1) + (y >> 1)) * (static_cast<unsigned>(W) >> 1) + (x >> 1)) * ucg + cg;
    const unsigned pos = (y & 1u) * 2u + (x & 1u);
    const u32x2 ib = reinterpret_cast<const u32x2*>(pool_idx)[wi];
    float dp[8];
    unpack8(reinterpret_cast<const u32x4*>(d_pooled)[wi], dp);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (((ib[e >> 2] >> (8 * (e & 3))) & 0xffu) == pos) g[e] += dp[e];
  }
}

// The grid stride is a multiple of CG (a power of two <= 256), so a thread keeps one channel octet.
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_bf16_kernel(const bf16_t* __restrict__ d_act,
                                                                      const bf16_t* __restrict__ y, const float* scale,
                                                                      const float* shift, const float* mean,
                                                                      const float* invstd, const bf16_t* d_pooled,
                                                                      const uint8_t* pool_idx, long items, int CG, int H,
                                                                      int W, float* __restrict__ partial) {
  __shared__ float sm[kThreads][17];
  const long first = blockIdx.x * static_cast<long>(kThreads) + threadIdx.x;
  const int c = static_cast<int>(first % CG) * 8;
  float sc[8], sh[8], mu[8], is[8], s1[8], s2[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sc[e] = scale[c + e];
    sh[e] = shift[c + e];
    mu[e] = mean[c + e];
    is[e] = invstd[c + e];
    s1[e] = 0.f;
    s2[e] = 0.f;
  }
  for (long i = first; i < items; i += static_cast<long>(gridDim.x) * kThreads) {
    float g[8], v[8];
    load_grad8(d_act, d_pooled, pool_idx, i, CG, H, W, g);
    unpack8(reinterpret_cast<const u32x4*>(y)[i], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float gg = (fmaf(v[e], sc[e], sh[e]) > 0.f) ? g[e] : 0.f;
      s1[e] += gg;
      s2[e] += gg * (v[e] - mu[e]) * is[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    sm[threadIdx.x][2 * e] = s1[e];
    sm[threadIdx.x][2 * e + 1] = s2[e];
  }
  __syncthreads();
  const int C = CG * 8;
  for (int cc = threadIdx.x; cc < C; cc += kThreads) {  // kThreads % CG == 0: thread t holds octet t % CG
    const int cg = cc >> 3, e = cc & 7;
    float a = 0.f, b = 0.f;
    for (int t = cg; t < kThreads; t += CG) {  // fixed order
      a += sm[t][2 * e];
      b += sm[t][2 * e + 1];
    }
    partial[(static_cast<long>(blockIdx.x) * C + cc) * 2 + 0] = a;
    partial[(static_cast<long>(blockIdx.x) * C + cc) * 2 + 1] = b;
  }
}

__global__ __launch_bounds__(kThreads) void bn_bwd_apply_bf16_kernel(const bf16_t* __restrict__ d_act,
                                                                     const bf16_t* __restrict__ y, const float* scale,
                                                                     const float* shift, const float* mean,
                                                                     const float* invstd, const float* gamma,
                                                                     const float* dgamma, const float* dbeta,
                                                                     const bf16_t* d_pooled, const uint8_t* pool_idx,
                                                                     float inv_count, long items, int CG, int H, int W,
                                                                     bf16_t* __restrict__ dy) {
  for (long i = blockIdx.x * static_cast<long>(kThreads) + threadIdx.x; i < items;
       i += static_cast<long>(gridDim.x) * kThreads) {
    const int c = static_cast<int>(static_cast<unsigned>(i) & static_cast<unsigned>(CG - 1)) * 8;  // CG is a power of two
    float g[8], v[8], out[8];
    load_grad8(d_act, d_pooled, pool_idx, i, CG, H, W, g);
    unpack8(reinterpret_cast<const u32x4*>(y)[i], v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      const float gg = (fmaf(v[e], scale[c + e], shift[c + e]) > 0.f) ? g[e] : 0.f;
      const float xhat = (v[e] - mean[c + e]) * invstd[c + e];
      out[e] = gamma[c + e] * invstd[c + e] * (gg - dbeta[c + e] * inv_count - xhat * dgamma[c + e] * inv_count);
    }
    reinterpret_cast<u32x4*>(dy)[i] = pack8(out);  // may alias d_act: the item was read by this thread above
  }
}

// ---- heads.  Forward: CG = C/8 lanes share a pixel (CG = 2^LOG2CG <= 16): every lane loads one octet, applies the
// dropout keep mask, multiplies it with the n_cls weight octets and the CG partial sums are folded across the lanes by
// DPP / ds_swizzle moves (xor_lane: the ds_bpermute shuffles of __shfl_xor made this kernel, like its fp32 twin,
// instruction bound at a third of the HBM rate).  DROP: 0 = none, 1 = counter hash, 2 = mask tensor -- separate
// instantiations keep the loop body straight-line.
template <int LOG2CG, int DROP, int PCLS>  // PCLS = classes padded to 4, 6 or 8 (5 key-point maps: configs[4]): the class loops carry no n_cls branches
__global__ __launch_bounds__(kThreads) void head_fwd_bf16_kernel(const bf16_t* __restrict__ x, const float* __restrict__ weight,
                                                                 const float* __restrict__ bias, long pixels, int HW,
                                                                 int n_cls, float keep_scale, uint32_t thr16,
                                                                 uint64_t seed, const uint8_t* __restrict__ mask, const uint64_t* __restrict__ seed_dev,
                                                                 float* __restrict__ out) {
  if (seed_dev != nullptr) seed += *seed_dev;  // graph-captured steps: the varying part of the seed lives in device memory
  constexpr int CG = 1 << LOG2CG, C = 8 * CG;
  __shared__ float wsm[PCLS * C];  // zero rows past n_cls (the class weights in registers ran 1.2x slower, twice measured)
  for (int i = threadIdx.x; i < PCLS * C; i += kThreads) wsm[i] = i < n_cls * C ? weight[i] : 0.f;
  __syncthreads();
  constexpr int ppb = kThreads >> LOG2CG;  // pixels per workgroup pass
  const int cg = threadIdx.x & (CG - 1), pl = threadIdx.x >> LOG2CG;
  const unsigned npix = static_cast<unsigned>(pixels), uhw = static_cast<unsigned>(HW);  // < 2^31 (launcher)
  const unsigned passes = (npix + ppb - 1) / ppb;
  for (unsigned ps = blockIdx.x; ps < passes; ps += gridDim.x) {  // all lanes stay in the loop: lane exchanges below
    const unsigned p = ps * ppb + pl;
    const bool live = p < npix;
    float f[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (live) {
      unpack8(reinterpret_cast<const u32x4*>(x)[(static_cast<long>(p) << LOG2CG) + cg], f);
      if constexpr (DROP == 1) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const uint64_t bits = keep_bits(seed, p, 2 * CG, 2 * cg + half);
#pragma unroll
          for (int q = 0; q < 4; ++q) f[4 * half + q] = keep_one(bits, q, thr16) ? f[4 * half + q] * keep_scale : 0.f;
        }
      } else if constexpr (DROP == 2) {
        const uint2 m8 = *reinterpret_cast<const uint2*>(mask + static_cast<long>(p) * C + cg * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const unsigned byte = ((e < 4 ? m8.x : m8.y) >> (8 * (e & 3))) & 0xffu;
          f[e] = byte != 0 ? f[e] * keep_scale : 0.f;
        }
      }
    }
    const unsigned n = p / uhw, hw = p - n * uhw;  // one 32-bit division per pixel
    float* obase = out + static_cast<long>(n) * n_cls * uhw + hw;
#pragma unroll
    for (int k = 0; k < PCLS; ++k) {
      float s = 0.f;
#pragma unroll
      for (int e = 0; e < 8; ++e) s = fmaf(f[e], wsm[k * C + cg * 8 + e], s);
      static_for<LOG2CG>([&](auto mc) { s += xor_lane<(1 << decltype(mc)::v)>(s); });  // every lane of the pixel holds the logit
      if (live && k < n_cls && (k & (CG - 1)) == cg)                // classes are dealt to the pixel's lanes round robin
        obase[static_cast<long>(k) * uhw] = 1.0f / (1.0f + __expf(-(s + bias[k])));
    }
  }
}

// Backward: 64-pixel tiles.  LDS: x*keep*scale fp32 [64][C+1], dlogit [64][8].  C = 8 * 2^LOG2CG: index arithmetic in
// shifts and 32 bits; a thread's octet position is the same for all its pieces, so its class weights stay in registers
// (the first version read 32 weights from LDS per octet); the (class, channel) pairs of the weight-gradient pass are
// decoded once; DROP (0 none, 1 counter hash, 2 mask tensor) keeps the piece loop free of per-element branches.
constexpr int kHeadTilePixels = 256;
template <int LOG2CG, int DROP, int PCLS>  // PCLS = classes padded to 4, 6 or 8 (5 key-point maps: configs[4]): the class loops carry no n_cls branches
__global__ __launch_bounds__(kThreads) void head_bwd_bf16_kernel(const float* __restrict__ d_out, const float* __restrict__ outp,
                                                                 const bf16_t* __restrict__ x, const float* __restrict__ weight,
                                                                 unsigned pixels, unsigned HW, int n_cls, float keep_scale,
                                                                 uint32_t thr16, uint64_t seed, const uint8_t* __restrict__ mask, const uint64_t* __restrict__ seed_dev,
                                                                 bf16_t* __restrict__ dx, int accumulate, int gate_x,
                                                                 float* __restrict__ partial, unsigned active) {
  if (seed_dev != nullptr) seed += *seed_dev;  // graph-captured steps: the varying part of the seed lives in device memory
  // 256-pixel tiles (four items in flight per thread at 32 channels): dlogit = d_out * out * (1 - out) of the tile goes through LDS (the NCHW class planes are read
  // coalesced along the pixels), then every thread takes (pixel, channel octet) items: dx = keep * scale * (W^T dlogit)
  // (+ old dx, ReLU gate of x) and the weight gradient of ITS octet, dlogit_k * (x * keep * scale), summed in registers
  // over all its items of the launch.  (Until round 3 the weight gradient went through LDS per tile -- x * keep * scale
  // written back, a third barrier and a 64-step loop of two LDS reads per (class, channel) pair: most of the kernel.)
  // One reduction at the end: lanes that share an octet by xor-shuffles, the four waves through LDS, fixed order.
  extern __shared__ float hsm[];
  constexpr int CG = 1 << LOG2CG, C = 8 * CG, TP = CG <= 8 ? kHeadTilePixels : 8 * kThreads / CG;  // <= 8 items per thread
  constexpr int ITEMS = (TP * CG + kThreads - 1) / kThreads;
  float* dl = hsm;                     // [TP][8]
  float* red = dl + TP * kHeadMaxCls;  // [4 waves][PCLS * C + PCLS]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int cg = tid & (CG - 1);
  const unsigned n_tiles = (pixels + TP - 1) / TP;
  // `active` workgroups walk the tiles (round 6).  A workgroup's fixed part -- 48 weight loads per thread, the shuffle and
  // LDS reduction of its 48 + 6 sums, a 1.3 KB row -- used to be paid per tile or two (one workgroup per tile up to 4096:
  // 2304 tiles at configs[4], 8192 at configs[3]); three workgroups per CU is what the registers allow to be resident.
  if (blockIdx.x >= n_tiles || blockIdx.x >= active) {
    // The grid and the partial rows are sized from 64-pixel tiles (unetpp_head_bwd_blocks, shared with the fp32 kernel);
    // blocks that own no tile of this kernel: a zero row (the caller sums every row) and out,
    // before the weight registers, the shuffles and the LDS reduction
    float* dst = partial + static_cast<long>(blockIdx.x) * (n_cls * C + n_cls);
    for (int i = tid; i < n_cls * C + n_cls; i += kThreads) dst[i] = 0.f;
    return;
  }
  float wq[PCLS][8], wacc[PCLS][8], bacc[PCLS];
#pragma unroll
  for (int k = 0; k < PCLS; ++k) {
    bacc[k] = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      wq[k][e] = (k < n_cls) ? weight[k * C + cg * 8 + e] : 0.f;
      wacc[k][e] = 0.f;
    }
  }
  for (int i = tid; i < TP * kHeadMaxCls; i += kThreads) dl[i] = 0.f;  // classes past n_cls are never written again
  for (unsigned tile = blockIdx.x; tile < n_tiles; tile += active) {
    const unsigned p0 = tile * TP;
    __syncthreads();
    for (int it = tid; it < TP * n_cls; it += kThreads) {  // dlogit = d_out * out * (1 - out)
      const int pl = it & (TP - 1), k = it / TP;
      const unsigned p = p0 + pl;
      float v = 0.f;
      if (p < pixels) {
        const unsigned n = p / HW, hw = p - n * HW;
        const long o = (static_cast<long>(n) * n_cls + k) * HW + hw;
        const float pr = outp[o];
        v = d_out[o] * pr * (1.f - pr);
      }
      dl[pl * kHeadMaxCls + k] = v;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < ITEMS; ++u) {
      const int it = tid + u * kThreads;
      if (ITEMS * kThreads != TP * CG && it >= TP * CG) break;
      const int pl = it >> LOG2CG;
      const unsigned p = p0 + pl;
      if (p >= pixels) continue;
      const long oct = (static_cast<long>(p) << LOG2CG) + cg;
      float raw[8], ks[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) ks[e] = 1.f;
      unpack8(reinterpret_cast<const u32x4*>(x)[oct], raw);
      if constexpr (DROP == 1) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          const uint64_t bits = keep_bits(seed, p, 2 * CG, 2 * cg + half);
#pragma unroll
          for (int q = 0; q < 4; ++q) ks[4 * half + q] = keep_one(bits, q, thr16) ? keep_scale : 0.f;
        }
      } else if constexpr (DROP == 2) {
        const uint2 m8 = *reinterpret_cast<const uint2*>(mask + oct * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) ks[e] = (((e < 4 ? m8.x : m8.y) >> (8 * (e & 3))) & 0xffu) != 0 ? keep_scale : 0.f;
      }
      float dk[PCLS];  // (rows past n_cls of dl are zero: written by the dlogit pass below n_cls only, cleared once)
#pragma unroll
      for (int k = 0; k < PCLS; ++k) dk[k] = dl[pl * kHeadMaxCls + k];
      float o[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float sum = 0.f;
#pragma unroll
        for (int k = 0; k < PCLS; ++k) sum = fmaf(wq[k][e], dk[k], sum);
        o[e] = sum * ks[e];
      }
      if (accumulate) {
        float old[8];
        unpack8(reinterpret_cast<const u32x4*>(dx)[oct], old);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] += old[e];
      }
      if (gate_x) {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = (raw[e] > 0.f) ? o[e] : 0.f;
      }
      reinterpret_cast<u32x4*>(dx)[oct] = pack8(o);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const float xd = raw[e] * ks[e];
#pragma unroll
        for (int k = 0; k < PCLS; ++k) wacc[k][e] = fmaf(dk[k], xd, wacc[k][e]);
      }
      if (cg == 0) {
#pragma unroll
        for (int k = 0; k < PCLS; ++k) bacc[k] += dk[k];
      }
    }
  }
  // ---- lanes of a wave that share an octet (lane bits >= LOG2CG), then the four waves: fixed order, reproducible ----
#pragma unroll
  for (int k = 0; k < PCLS; ++k) {
#pragma unroll
    for (int m = CG; m < 64; m <<= 1) bacc[k] += __shfl_xor(bacc[k], m);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      float v = wacc[k][e];
#pragma unroll
      for (int m = CG; m < 64; m <<= 1) v += __shfl_xor(v, m);
      wacc[k][e] = v;
    }
  }
  __syncthreads();
  constexpr int ROW = PCLS * C + PCLS;
  if (lane < CG) {
#pragma unroll
    for (int k = 0; k < PCLS; ++k)
#pragma unroll
      for (int e = 0; e < 8; ++e) red[wave * ROW + k * C + cg * 8 + e] = wacc[k][e];
  }
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < PCLS; ++k) red[wave * ROW + PCLS * C + k] = bacc[k];
  }
  __syncthreads();
  const int NW = n_cls * C;
  float* dst = partial + static_cast<long>(blockIdx.x) * (NW + n_cls);
  for (int i = tid; i < NW; i += kThreads) dst[i] = (red[i] + red[ROW + i]) + (red[2 * ROW + i] + red[3 * ROW + i]);
  if (tid < n_cls)
    dst[NW + tid] = (red[PCLS * C + tid] + red[ROW + PCLS * C + tid]) + (red[2 * ROW + PCLS * C + tid] + red[3 * ROW + PCLS * C + tid]);
}

inline bool octets_ok(int C) {  // C = 8 * CG, CG a power of two <= 256 (a thread keeps its octet across a grid stride)
  const int cg = C >> 3;
  return C >= 8 && (C & 7) == 0 && (cg & (cg - 1)) == 0 && cg <= 256;
}


// ---- max-pool backward without a BatchNorm to route through (is_batchnorm=False): thread = (pixel, octet) ----
__global__ __launch_bounds__(kThreads) void maxpool_bwd_bf16_kernel(const bf16_t* __restrict__ d_pooled,
                                                                    const uint8_t* __restrict__ pool_idx, int N, int H, int W,
                                                                    int CG, bf16_t* __restrict__ d_act,
                                                                    const bf16_t* __restrict__ gate) {
  const long items = static_cast<long>(N) * H * W * CG;
  const unsigned ucg = static_cast<unsigned>(CG);
  for (long i = blockIdx.x * static_cast<long>(kThreads) + threadIdx.x; i < items;
       i += static_cast<long>(gridDim.x) * kThreads) {
    const unsigned iu = static_cast<unsigned>(i);  // items < 2^31 (launcher)
    const unsigned cg = iu % ucg;
    unsigned r = iu / ucg;
    const unsigned x = r % static_cast<unsigned>(W);
    r /= static_cast<unsigned>(W);
    const unsigned y = r % static_cast<unsigned>(H);
    const unsigned n = r / static_cast<unsigned>(H);
    const unsigned wi = ((n * (static_cast<unsigned>(H) >> 1) + (y >> 1)) * (static_cast<unsigned>(W) >> 1) + (x >> 1)) * ucg + cg;
    const unsigned pos = (y & 1u) * 2u + (x & 1u);
    const u32x2 ib = reinterpret_cast<const u32x2*>(pool_idx)[wi];
    float g[8], dp[8];
    unpack8(reinterpret_cast<const u32x4*>(d_act)[i], g);
    unpack8(reinterpret_cast<const u32x4*>(d_pooled)[wi], dp);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      if (((ib[e >> 2] >> (8 * (e & 3))) & 0xffu) == pos) g[e] += dp[e];
    if (gate != nullptr) {
      float gt[8];
      unpack8(reinterpret_cast<const u32x4*>(gate)[i], gt);
#pragma unroll
      for (int e = 0; e < 8; ++e) g[e] = (gt[e] > 0.f) ? g[e] : 0.f;
    }
    reinterpret_cast<u32x4*>(d_act)[i] = pack8(g);
  }
}

// ---- bilinear x2, align_corners=True (nn.UpsamplingBilinear2d, models/unet.py:190) on bf16 NHWC ----
__device__ __forceinline__ void bilinear_src_bf(int dst, int in_size, float rscale, int& i0, int& i1, float& l1) {
  const float s = rscale * static_cast<float>(dst);
  i0 = static_cast<int>(s);
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = s - static_cast<float>(i0);
}

// thread = (output pixel, octet): four 16-byte loads, the fp32 expression of the fp32 kernel, one rounding
__global__ __launch_bounds__(kThreads) void bilinear2x_fwd_bf16_kernel(const bf16_t* __restrict__ x, int N, int H, int W,
                                                                       int CG, bf16_t* __restrict__ y) {
  const int Ho = 2 * H, Wo = 2 * W;
  const float ry = (Ho > 1) ? static_cast<float>(H - 1) / static_cast<float>(Ho - 1) : 0.f;
  const float rx = (Wo > 1) ? static_cast<float>(W - 1) / static_cast<float>(Wo - 1) : 0.f;
  const long items = static_cast<long>(N) * Ho * Wo * CG;
  for (long i = blockIdx.x * static_cast<long>(kThreads) + threadIdx.x; i < items;
       i += static_cast<long>(gridDim.x) * kThreads) {
    const int cg = static_cast<int>(i % CG);
    long r = i / CG;
    const int xo = static_cast<int>(r % Wo);
    r /= Wo;
    const int yo = static_cast<int>(r % Ho);
    const long n = r / Ho;
    int y0, y1, x0, x1;
    float ly, lx;
    bilinear_src_bf(yo, H, ry, y0, y1, ly);
    bilinear_src_bf(xo, W, rx, x0, x1, lx);
    const u32x4* b = reinterpret_cast<const u32x4*>(x) + n * H * W * CG + cg;
    float v00[8], v01[8], v10[8], v11[8], o[8];
    unpack8(b[(static_cast<long>(y0) * W + x0) * CG], v00);
    unpack8(b[(static_cast<long>(y0) * W + x1) * CG], v01);
    unpack8(b[(static_cast<long>(y1) * W + x0) * CG], v10);
    unpack8(b[(static_cast<long>(y1) * W + x1) * CG], v11);
#pragma unroll
    for (int e = 0; e < 8; ++e)
      o[e] = (1.f - ly) * ((1.f - lx) * v00[e] + lx * v01[e]) + ly * ((1.f - lx) * v10[e] + lx * v11[e]);
    reinterpret_cast<u32x4*>(y)[i] = pack8(o);
  }
}

// gather form of the transposed stencil (fixed summation order, no atomics): thread = (source pixel, octet)
__global__ __launch_bounds__(kThreads) void bilinear2x_bwd_bf16_kernel(const bf16_t* __restrict__ dy, int N, int H, int W,
                                                                       int CG, bf16_t* __restrict__ dx, int accumulate,
                                                                       const bf16_t* __restrict__ gate) {
  const int Ho = 2 * H, Wo = 2 * W;
  const float ry = (Ho > 1) ? static_cast<float>(H - 1) / static_cast<float>(Ho - 1) : 0.f;
  const float rx = (Wo > 1) ? static_cast<float>(W - 1) / static_cast<float>(Wo - 1) : 0.f;
  const long items = static_cast<long>(N) * H * W * CG;
  for (long i = blockIdx.x * static_cast<long>(kThreads) + threadIdx.x; i < items;
       i += static_cast<long>(gridDim.x) * kThreads) {
    const int cg = static_cast<int>(i % CG);
    long r = i / CG;
    const int xs = static_cast<int>(r % W);
    r /= W;
    const int ys = static_cast<int>(r % H);
    const long n = r / H;
    const int ylo = max(0, 2 * ys - 3), yhi = min(Ho - 1, 2 * ys + 3);
    const int xlo = max(0, 2 * xs - 3), xhi = min(Wo - 1, 2 * xs + 3);
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int yo = ylo; yo <= yhi; ++yo) {
      int y0, y1;
      float ly;
      bilinear_src_bf(yo, H, ry, y0, y1, ly);
      float wy = 0.f;
      if (y0 == ys) wy += 1.f - ly;
      if (y1 == ys) wy += ly;
      if (wy == 0.f) continue;
      for (int xo = xlo; xo <= xhi; ++xo) {
        int x0, x1;
        float lx;
        bilinear_src_bf(xo, W, rx, x0, x1, lx);
        float wx = 0.f;
        if (x0 == xs) wx += 1.f - lx;
        if (x1 == xs) wx += lx;
        if (wx != 0.f) {
          float v[8];
          unpack8(reinterpret_cast<const u32x4*>(dy)[((n * Ho + yo) * Wo + xo) * CG + cg], v);
#pragma unroll
          for (int e = 0; e < 8; ++e) s[e] += wy * wx * v[e];
        }
      }
    }
    if (accumulate) {
      float old[8];
      unpack8(reinterpret_cast<const u32x4*>(dx)[i], old);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += old[e];
    }
    if (gate != nullptr) {
      float gt[8];
      unpack8(reinterpret_cast<const u32x4*>(gate)[i], gt);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] = (gt[e] > 0.f) ? s[e] : 0.f;
    }
    reinterpret_cast<u32x4*>(dx)[i] = pack8(s);
  }
}

}  // namespace
}  // namespace unetpp

using namespace unetpp;

extern "C" int unetpp_affine_relu_pool_bf16(const void* y, const float* scale, const float* shift, int32_t relu, int32_t N,
                                            int32_t H, int32_t W, int32_t C, void* act, void* pooled, uint8_t* pool_idx,
                                            void* stream) {
  if (!y || N < 1 || H < 1 || W < 1 || C < 8 || (C & 7) || !a16(y) || (act && !a16(act))) return UNETPP_EINVAL;
  if ((scale == nullptr) != (shift == nullptr) || (act == nullptr && pooled == nullptr)) return UNETPP_EINVAL;
  const int CG = C >> 3;
  if (pooled == nullptr) {
    const long items = static_cast<long>(N) * H * W * CG;
    hipLaunchKernelGGL(affine_relu_bf16_kernel, dim3(grid_for8(items)), dim3(kThreads), 0, ST(stream),
                       static_cast<const bf16_t*>(y), scale, shift, relu, items, CG, static_cast<bf16_t*>(act));
    return launch_status();
  }
  if ((H & 1) || (W & 1) || !pool_idx || !a16(pooled) || (reinterpret_cast<uintptr_t>(pool_idx) & 7)) return UNETPP_EINVAL;
  const long items = static_cast<long>(N) * (H / 2) * (W / 2) * CG;
  if (items >= 0x7fffffffL) return UNETPP_EINVAL;
  hipLaunchKernelGGL(affine_relu_pool_bf16_kernel, dim3(grid_for8(items)), dim3(kThreads), 0, ST(stream),
                     static_cast<const bf16_t*>(y), scale, shift, relu, N, H, W, CG, static_cast<bf16_t*>(act),
                     static_cast<bf16_t*>(pooled), pool_idx);
  return launch_status();
}

extern "C" int64_t unetpp_bn_bwd_blocks_bf16(int64_t pixels, int32_t C) {
  if (pixels < 1 || !octets_ok(C)) return 0;
  const long items = pixels * (C >> 3);
  long b = (items + 4L * kThreads - 1) / (4L * kThreads);  // >= 4 items per thread
  if (b > 2048) b = 2048;
  return b < 1 ? 1 : b;
}

extern "C" int unetpp_bn_bwd_reduce_bf16(const void* d_act, const void* y, const float* scale, const float* shift,
                                         const float* mean, const float* invstd, const void* d_pooled,
                                         const uint8_t* pool_idx, int32_t N, int32_t H, int32_t W, int32_t C, float* partial,
                                         void* stream) {
  if (!d_act || !y || !scale || !shift || !mean || !invstd || !partial || N < 1 || H < 1 || W < 1 || !octets_ok(C))
    return UNETPP_EINVAL;
  if ((d_pooled == nullptr) != (pool_idx == nullptr) || (d_pooled != nullptr && ((H | W) & 1))) return UNETPP_EINVAL;
  const long pixels = static_cast<long>(N) * H * W;
  if (pixels * (C >> 3) >= 0x7fffffffL) return UNETPP_EINVAL;
  hipLaunchKernelGGL(bn_bwd_reduce_bf16_kernel, dim3(static_cast<unsigned>(unetpp_bn_bwd_blocks_bf16(pixels, C))),
                     dim3(kThreads), 0, ST(stream), static_cast<const bf16_t*>(d_act), static_cast<const bf16_t*>(y), scale,
                     shift, mean, invstd, static_cast<const bf16_t*>(d_pooled), pool_idx, pixels * (C >> 3), C >> 3, H, W,
                     partial);
  return launch_status();
}

extern "C" int unetpp_bn_bwd_apply_bf16(const void* d_act, const void* y, const float* scale, const float* shift,
                                        const float* mean, const float* invstd, const float* gamma, const float* dgamma,
                                        const float* dbeta, const void* d_pooled, const uint8_t* pool_idx, int32_t N,
                                        int32_t H, int32_t W, int32_t C, void* dy, void* stream) {
  if (!d_act || !y || !scale || !shift || !mean || !invstd || !gamma || !dgamma || !dbeta || !dy || N < 1 || H < 1 ||
      W < 1 || !octets_ok(C))
    return UNETPP_EINVAL;
  if ((d_pooled == nullptr) != (pool_idx == nullptr) || (d_pooled != nullptr && ((H | W) & 1))) return UNETPP_EINVAL;
  const long pixels = static_cast<long>(N) * H * W;
  const long items = pixels * (C >> 3);
  if (items >= 0x7fffffffL) return UNETPP_EINVAL;
  hipLaunchKernelGGL(bn_bwd_apply_bf16_kernel, dim3(grid_for8(items)), dim3(kThreads), 0, ST(stream),
                     static_cast<const bf16_t*>(d_act), static_cast<const bf16_t*>(y), scale, shift, mean, invstd, gamma,
                     dgamma, dbeta, static_cast<const bf16_t*>(d_pooled), pool_idx, 1.0f / static_cast<float>(pixels), items,
                     C >> 3, H, W, static_cast<bf16_t*>(dy));
  return launch_status();
}

static bool head_bf16_ok(int N, int H, int W, int C, int n_cls, float p_drop) {
  const int cg = C >> 3;
  return N >= 1 && H >= 1 && W >= 1 && C >= 8 && (C & 7) == 0 && C <= kHeadMaxC && (cg & (cg - 1)) == 0 && n_cls >= 1 &&
         n_cls <= kHeadMaxCls && p_drop >= 0.f && p_drop < 1.f;
}

extern "C" int unetpp_head_fwd_bf16(const void* x, const float* weight, const float* bias, int32_t N, int32_t H, int32_t W,
                                    int32_t C, int32_t n_cls, float p_drop, uint64_t seed, const uint8_t* mask, const uint64_t* seed_dev,
                                    float* out_nchw, void* stream) {
  if (!x || !weight || !bias || !out_nchw || !a16(x) || !head_bf16_ok(N, H, W, C, n_cls, p_drop)) return UNETPP_EINVAL;
  const long pixels = static_cast<long>(N) * H * W;
  if (pixels >= 0x7fffffffL) return UNETPP_EINVAL;
  const int CG = C >> 3;
  const long passes = (pixels + kThreads / CG - 1) / (kThreads / CG);
  const dim3 grid(static_cast<unsigned>(passes < 256 * 16 ? passes : 256 * 16));
  const int drop = p_drop > 0.f ? (mask == nullptr ? 1 : 2) : 0;
  if (drop == 2 && (reinterpret_cast<uintptr_t>(mask) & 7) != 0) return UNETPP_EINVAL;  // mask octets are read as 8 bytes
#define UNETPP_HEAD_BF(L, D, PC)                                                                                          \
  hipLaunchKernelGGL((head_fwd_bf16_kernel<L, D, PC>), grid, dim3(kThreads), 0, ST(stream), static_cast<const bf16_t*>(x), \
                     weight, bias, pixels, H * W, n_cls, 1.0f / (1.0f - p_drop), keep_threshold(p_drop), seed, mask, seed_dev,       \
                     out_nchw)
#define UNETPP_HEAD_BF_D(L, D)            \
  do {                                    \
    if (n_cls <= 4) UNETPP_HEAD_BF(L, D, 4); \
    else if (n_cls <= 6) UNETPP_HEAD_BF(L, D, 6); \
    else UNETPP_HEAD_BF(L, D, 8);         \
  } while (0)
#define UNETPP_HEAD_BF_L(L)              \
  do {                                   \
    if (drop == 0) UNETPP_HEAD_BF_D(L, 0); \
    else if (drop == 1) UNETPP_HEAD_BF_D(L, 1); \
    else UNETPP_HEAD_BF_D(L, 2);         \
  } while (0)
  switch (CG) {
    case 1: UNETPP_HEAD_BF_L(0); break;
    case 2: UNETPP_HEAD_BF_L(1); break;
    case 4: UNETPP_HEAD_BF_L(2); break;
    case 8: UNETPP_HEAD_BF_L(3); break;
    default: UNETPP_HEAD_BF_L(4); break;
  }
#undef UNETPP_HEAD_BF_L
#undef UNETPP_HEAD_BF_D
#undef UNETPP_HEAD_BF
  return launch_status();
}

extern "C" int unetpp_head_bwd_bf16(const float* d_out_nchw, const float* out_nchw, const void* x, const float* weight,
                                    int32_t N, int32_t H, int32_t W, int32_t C, int32_t n_cls, float p_drop, uint64_t seed,
                                    const uint8_t* mask, const uint64_t* seed_dev, void* dx, int32_t accumulate, int32_t gate_x, float* partial,
                                    void* stream) {
  if (!d_out_nchw || !out_nchw || !x || !weight || !dx || !partial || !a16(x) || !a16(dx) ||
      !head_bf16_ok(N, H, W, C, n_cls, p_drop))
    return UNETPP_EINVAL;
  const long pixels = static_cast<long>(N) * H * W;
  if (pixels >= 0x7fffffffL) return UNETPP_EINVAL;
  const long tiles = (pixels + 63) / 64;
  const size_t lds = (kHeadTilePixels * kHeadMaxCls + 4 * (kHeadMaxCls * C + kHeadMaxCls)) * sizeof(float);  // dlogit tile + 4 wave rows
  const dim3 grid(static_cast<unsigned>(tiles < 4096 ? tiles : 4096));
  // Workgroups that take tiles: at most HEAD_WGS_PER_CU per CU (default 4), and then as few as walk the same number of
  // rounds (2304 tiles: 3 rounds of 768 rather than 1024 workgroups of which 256 carry a third tile).  tools/sweep_head_wgs.sh:
  // configs[4] 87 -> 65-74 us per head, configs[3] 113 -> 102 us (dropout, accumulate and gate on; 0 = every workgroup).
  const long per_cu = opt_value(OPT_HEAD_WGS_PER_CU, 4);
  const int cus = device_cu_count();
  if (cus <= 0) return UNETPP_ELAUNCH;
  unsigned active = grid.x;
  {
    constexpr long kTilePixels = kHeadTilePixels;  // (every instantiation with C <= 64; wider heads use smaller tiles: more rounds, same rule)
    const long n_tiles = (pixels + kTilePixels - 1) / kTilePixels;
    const long most = per_cu * cus;
    if (per_cu > 0 && most < n_tiles) {
      const long rounds = (n_tiles + most - 1) / most;
      active = static_cast<unsigned>((n_tiles + rounds - 1) / rounds);
    }
    if (active > grid.x) active = grid.x;
  }
  const int drop = p_drop > 0.f ? (mask == nullptr ? 1 : 2) : 0;
  if (drop == 2 && (reinterpret_cast<uintptr_t>(mask) & 7) != 0) return UNETPP_EINVAL;  // mask octets are read as 8 bytes
#define UNETPP_HEAD_BWD_BF(L, D)                                    \
  do {                                                              \
    if (n_cls <= 4) UNETPP_HEAD_BWD_BF_P(L, D, 4);                  \
    else if (n_cls <= 6) UNETPP_HEAD_BWD_BF_P(L, D, 6);             \
    else UNETPP_HEAD_BWD_BF_P(L, D, 8);                             \
  } while (0)
#define UNETPP_HEAD_BWD_BF_P(L, D, PC)                                                                                 \
  hipLaunchKernelGGL((head_bwd_bf16_kernel<L, D, PC>), grid, dim3(kThreads), lds, ST(stream), d_out_nchw, out_nchw,     \
                     static_cast<const bf16_t*>(x), weight, static_cast<unsigned>(pixels), static_cast<unsigned>(H * W), \
                     n_cls, 1.0f / (1.0f - p_drop), keep_threshold(p_drop), seed, mask, seed_dev, static_cast<bf16_t*>(dx),       \
                     accumulate, gate_x, partial, active)
#define UNETPP_HEAD_BWD_BF_L(L)              \
  do {                                       \
    if (drop == 0) UNETPP_HEAD_BWD_BF(L, 0); \
    else if (drop == 1) UNETPP_HEAD_BWD_BF(L, 1); \
    else UNETPP_HEAD_BWD_BF(L, 2);           \
  } while (0)
  switch (C >> 3) {
    case 1: UNETPP_HEAD_BWD_BF_L(0); break;
    case 2: UNETPP_HEAD_BWD_BF_L(1); break;
    case 4: UNETPP_HEAD_BWD_BF_L(2); break;
    case 8: UNETPP_HEAD_BWD_BF_L(3); break;
    default: UNETPP_HEAD_BWD_BF_L(4); break;
  }
#undef UNETPP_HEAD_BWD_BF_L
#undef UNETPP_HEAD_BWD_BF
#undef UNETPP_HEAD_BWD_BF_P
  return launch_status();
}

extern "C" int unetpp_maxpool_bwd_bf16(const void* d_pooled, const uint8_t* pool_idx, int32_t N, int32_t H, int32_t W,
                                       int32_t C, void* d_act, const void* gate, void* stream) {
  if (!d_pooled || !pool_idx || !d_act || N < 1 || H < 2 || W < 2 || (H & 1) || (W & 1) || C < 8 || (C & 7)) return UNETPP_EINVAL;
  if (!a16(d_pooled) || !a16(d_act) || (gate && !a16(gate)) || (reinterpret_cast<uintptr_t>(pool_idx) & 7)) return UNETPP_EINVAL;
  const long items = static_cast<long>(N) * H * W * (C >> 3);
  if (items >= 0x7fffffffL) return UNETPP_EINVAL;
  hipLaunchKernelGGL(maxpool_bwd_bf16_kernel, dim3(grid_for8(items)), dim3(kThreads), 0, ST(stream),
                     static_cast<const bf16_t*>(d_pooled), pool_idx, N, H, W, C >> 3, static_cast<bf16_t*>(d_act),
                     static_cast<const bf16_t*>(gate));
  return launch_status();
}

extern "C" int unetpp_bilinear2x_fwd_bf16(const void* x, int32_t N, int32_t H, int32_t W, int32_t C, void* y, void* stream) {
  if (!x || !y || N < 1 || H < 1 || W < 1 || C < 8 || (C & 7) || !a16(x) || !a16(y)) return UNETPP_EINVAL;
  const long items = static_cast<long>(N) * 4 * H * W * (C >> 3);
  hipLaunchKernelGGL(bilinear2x_fwd_bf16_kernel, dim3(grid_for8(items)), dim3(kThreads), 0, ST(stream),
                     static_cast<const bf16_t*>(x), N, H, W, C >> 3, static_cast<bf16_t*>(y));
  return launch_status();
}

extern "C" int unetpp_bilinear2x_bwd_bf16(const void* dy, int32_t N, int32_t H, int32_t W, int32_t C, void* dx,
                                          int32_t accumulate, const void* gate, void* stream) {
  if (!dy || !dx || N < 1 || H < 1 || W < 1 || C < 8 || (C & 7) || !a16(dy) || !a16(dx) || (gate && !a16(gate))) return UNETPP_EINVAL;
  const long items = static_cast<long>(N) * H * W * (C >> 3);
  hipLaunchKernelGGL(bilinear2x_bwd_bf16_kernel, dim3(grid_for8(items)), dim3(kThreads), 0, ST(stream),
                     static_cast<const bf16_t*>(dy), N, H, W, C >> 3, static_cast<bf16_t*>(dx), accumulate,
                     static_cast<const bf16_t*>(gate));
  return launch_status();
}

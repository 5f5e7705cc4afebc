// HBM-bound companions of the MFMA kernels: BatchNorm statistics / apply / backward, fused
// affine+ReLU+2x2 max-pool, the 1x1 sigmoid heads with dropout, bilinear x2 (align_corners),
// and the NCHW<->NHWC converters used at the network edge.  All NHWC fp32; 16-byte vector
// accesses whenever the channel count is a multiple of 4, scalar fallback otherwise.
#include "common.h"
#include "lds_asm.h"
#include "dropout.h"

namespace unetpp {
namespace {

template <int VEC>
struct Pack;
template <>
struct Pack<4> {
  using T = f32x4;
};
template <>
struct Pack<1> {
  using T = float;
};
template <int VEC>
__device__ __forceinline__ float& lane_of(typename Pack<VEC>::T& v, int i) {
  if constexpr (VEC == 4)
    return reinterpret_cast<float*>(&v)[i];
  else
    return v;
}
template <int VEC>
__device__ __forceinline__ float lane_of(const typename Pack<VEC>::T& v, int i) {
  if constexpr (VEC == 4)
    return v[i];
  else
    return v;
}

inline unsigned grid_for(long items, int cap = 2048 * 8) {
  long b = (items + kThreads - 1) / kThreads;
  if (b < 1) b = 1;
  if (b > cap) b = cap;
  return static_cast<unsigned>(b);
}

// ------------------------------------------------------------------ BatchNorm statistics
// partial [n_blocks][C][2] -> per-channel totals in double; one workgroup (kFinThreads or fewer threads) per channel:
// 8-byte loads, eight rows in flight per thread, wave shuffles, then the waves' totals in wave order.
constexpr int kFinThreads = 1024;
__device__ __forceinline__ void reduce_pair_over_blocks(const float* partial, long n_blocks, int C, int c,
                                                         double& s1, double& s2) {
  __shared__ double red[2][kFinThreads / 64];
  double a = 0.0, b = 0.0;
  const float2* p = reinterpret_cast<const float2*>(partial) + c;
  const long step = blockDim.x;
  long i = threadIdx.x;
  for (; i + 7 * step < n_blocks; i += 8 * step) {
    float2 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = p[(i + u * step) * C];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a += static_cast<double>(v[u].x);
      b += static_cast<double>(v[u].y);
    }
  }
  for (; i < n_blocks; i += step) {
    const float2 v = p[i * C];
    a += static_cast<double>(v.x);
    b += static_cast<double>(v.y);
  }
#pragma unroll
  for (int m = 32; m > 0; m >>= 1) {
    a += __shfl_xor(a, m);
    b += __shfl_xor(b, m);
  }
  const int wave = threadIdx.x >> 6, n_waves = blockDim.x >> 6;
  if ((threadIdx.x & 63) == 0) {
    red[0][wave] = a;
    red[1][wave] = b;
  }
  __syncthreads();
  a = 0.0;
  b = 0.0;
  for (int w = 0; w < n_waves; ++w) {
    a += red[0][w];
    b += red[1][w];
  }
  s1 = a;
  s2 = b;
}

__global__ __launch_bounds__(kFinThreads) void bn_finalize_kernel(const float* partial, long n_blocks, int C, long count,
                                                               const float* gamma, const float* beta, float eps,
                                                               float momentum, float* running_mean, float* running_var,
                                                               float* mean, float* invstd, float* scale, float* shift) {
  const int c = blockIdx.x;
  double s1, s2;
  reduce_pair_over_blocks(partial, n_blocks, C, c, s1, s2);
  if (threadIdx.x == 0) {
    const double m = s1 / static_cast<double>(count);
    double var = s2 / static_cast<double>(count) - m * m;
    if (var < 0.0) var = 0.0;
    const double is = 1.0 / sqrt(var + static_cast<double>(eps));
    const double sc = static_cast<double>(gamma[c]) * is;
    mean[c] = static_cast<float>(m);
    invstd[c] = static_cast<float>(is);
    scale[c] = static_cast<float>(sc);
    shift[c] = static_cast<float>(static_cast<double>(beta[c]) - m * sc);
    if (running_mean != nullptr) {
      const double unbiased = count > 1 ? var * static_cast<double>(count) / static_cast<double>(count - 1) : var;
      running_mean[c] = static_cast<float>((1.0 - momentum) * running_mean[c] + momentum * m);
      running_var[c] = static_cast<float>((1.0 - momentum) * running_var[c] + momentum * unbiased);
    }
  }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm, const float* rv, float eps,
                                      int C, float* scale, float* shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float is = 1.0f / sqrtf(rv[c] + eps);
  const float sc = gamma[c] * is;
  scale[c] = sc;
  shift[c] = beta[c] - rm[c] * sc;
}

__global__ __launch_bounds__(kFinThreads) void bn_bwd_finalize_kernel(const float* partial, long n_blocks, int C,
                                                                   float* dgamma, float* dbeta) {
  const int c = blockIdx.x;
  double s1, s2;
  reduce_pair_over_blocks(partial, n_blocks, C, c, s1, s2);
  if (threadIdx.x == 0) {
    dbeta[c] = static_cast<float>(s1);
    dgamma[c] = static_cast<float>(s2);
  }
}

// ------------------------------------------------------------------ affine + ReLU (+ 2x2 max-pool)
template <int VEC>
__global__ __launch_bounds__(kThreads) void affine_relu_kernel(const float* __restrict__ y, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, int relu, long items,
                                                               int CG, float* __restrict__ act) {
  using P = typename Pack<VEC>::T;
  for (long i = blockIdx.x * static_cast<long>(kThreads) + threadIdx.x; i < items;
       i += static_cast<long>(gridDim.x) * kThreads) {
    const int c = static_cast<int>(i % CG) * VEC;
    P v = reinterpret_cast<const P*>(y)[i];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      float t = lane_of<VEC>(v, k);
      if (scale != nullptr) t = fmaf(t, scale[c + k], shift[c + k]);
      if (relu) t = fmaxf(t, 0.f);
      lane_of<VEC>(v, k) = t;
    }
    reinterpret_cast<P*>(act)[i] = v;
  }
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void affine_relu_pool_kernel(const float* __restrict__ y,
                                                                    const float* __restrict__ scale,
                                                                    const float* __restrict__ shift, int relu, int N,
                                                                    int H, int W, int CG, float* __restrict__ act,
                                                                    float* __restrict__ pooled,
                                                                    uint8_t* __restrict__ pool_idx) {
  using P = typename Pack<VEC>::T;
  const int Ho = H >> 1, Wo = W >> 1;
  const long items = static_cast<long>(N) * Ho * Wo * CG;
  for (long i = blockIdx.x * static_cast<long>(kThreads) + threadIdx.x; i < items;
       i += static_cast<long>(gridDim.x) * kThreads) {
    const int cg = static_cast<int>(i % CG);
    long r = i / CG;
    const int xo = static_cast<int>(r % Wo);
    r /= Wo;
    const int yo = static_cast<int>(r % Ho);
    const int n = static_cast<int>(r / Ho);
    const int c = cg * VEC;
    float best[VEC];
    uint8_t bi[VEC];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const long pix = (static_cast<long>(n) * H + (2 * yo + (q >> 1))) * W + (2 * xo + (q & 1));
      P v = reinterpret_cast<const P*>(y)[pix * CG + cg];
#pragma unroll
      for (int k = 0; k < VEC; ++k) {
        float t = lane_of<VEC>(v, k);
        if (scale != nullptr) t = fmaf(t, scale[c + k], shift[c + k]);
        if (relu) t = fmaxf(t, 0.f);
        lane_of<VEC>(v, k) = t;
        if (q == 0 || t > best[k]) {  // first maximum wins, scan order (0,0),(0,1),(1,0),(1,1)
          best[k] = t;
          bi[k] = static_cast<uint8_t>(q);
        }
      }
      if (act != nullptr) reinterpret_cast<P*>(act)[pix * CG + cg] = v;
    }
    P out;
#pragma unroll
    for (int k = 0; k < VEC; ++k) lane_of<VEC>(out, k) = best[k];
    reinterpret_cast<P*>(pooled)[i] = out;
    if (pool_idx != nullptr) {
#pragma unroll
      for (int k = 0; k < VEC; ++k) pool_idx[i * VEC + k] = bi[k];
    }
  }
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void maxpool_bwd_kernel(const float* __restrict__ d_pooled,
                                                               const uint8_t* __restrict__ pool_idx, int N, int H, int W,
                                                               int CG, float* __restrict__ d_act) {
  const int Ho = H >> 1, Wo = W >> 1;
  const long items = static_cast<long>(N) * Ho * Wo * CG;
  for (long i = blockIdx.x * static_cast<long>(kThreads) + threadIdx.x; i < items;
       i += static_cast<long>(gridDim.x) * kThreads) {
    const int cg = static_cast<int>(i % CG);
    long r = i / CG;
    const int xo = static_cast<int>(r % Wo);
    r /= Wo;
    const int yo = static_cast<int>(r % Ho);
    const int n = static_cast<int>(r / Ho);
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const int q = pool_idx[i * VEC + k];
      const long pix = (static_cast<long>(n) * H + (2 * yo + (q >> 1))) * W + (2 * xo + (q & 1));
      d_act[(pix * CG + cg) * VEC + k] += d_pooled[i * VEC + k];
    }
  }
}

// Row-structured forms of the two pool kernels for 4-aligned channels and tensors below 2^31 elements: a workgroup
// walks whole rows of windows (row = n * Ho + yo), a thread the (window, channel quad) items j = xo * CG + cg of the
// row, so the index arithmetic is 32-bit and division free (the item's two input rows sit at 2j - cg and
// 2j - cg + CG quads from the row starts) and the four argmax bytes of a quad leave as one dword.
// Streaming (non-temporal) accesses for the two full-resolution tensors of the pool pass (y is not read again before
// the backward pass, act not before the decoder): 114 -> 98 us on the X_0,0 shape.  -DUNETPP_POOL_NO_NT turns them off.
#ifndef UNETPP_POOL_NO_NT
#define UNETPP_POOL_LOAD(p) __builtin_nontemporal_load(p)
#define UNETPP_POOL_STORE(p, v) __builtin_nontemporal_store(v, p)
#else
#define UNETPP_POOL_LOAD(p) (*(p))
#define UNETPP_POOL_STORE(p, v) (*(p) = (v))
#endif
__global__ __launch_bounds__(kThreads) void affine_relu_pool_rows_kernel(
    const f32x4* __restrict__ y, const f32x4* __restrict__ scale, const f32x4* __restrict__ shift, int relu,
    unsigned rows, unsigned Wo, unsigned CG, f32x4* __restrict__ act, f32x4* __restrict__ pooled,
    uint32_t* __restrict__ pool_idx) {
  const unsigned row_items = Wo * CG;  // quads per pooled row; an input row holds 2 * row_items quads
  for (unsigned row = blockIdx.x; row < rows; row += gridDim.x) {
    const unsigned in0 = row * 4u * row_items;  // input row 2 * row (rows of one image are consecutive: H = 2 Ho)
    for (unsigned j = threadIdx.x; j < row_items; j += kThreads) {
      const unsigned cg = j % CG;
      const unsigned a = in0 + 2u * j - cg;
      const unsigned off[4] = {a, a + CG, a + 2u * row_items, a + 2u * row_items + CG};
      f32x4 v[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) v[q] = UNETPP_POOL_LOAD(y + off[q]);
      f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
      if (scale != nullptr) {
        sc = scale[cg];
        sh = shift[cg];
      }
      f32x4 best;
      uint32_t bi = 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          float t = fmaf(v[q][k], sc[k], sh[k]);
          if (relu) t = fmaxf(t, 0.f);
          v[q][k] = t;
          if (q == 0) {
            best[k] = t;
          } else if (t > best[k]) {  // first maximum wins, scan order (0,0),(0,1),(1,0),(1,1)
            best[k] = t;
            bi = (bi & ~(0xffu << (8 * k))) | (static_cast<uint32_t>(q) << (8 * k));
          }
        }
        if (act != nullptr) UNETPP_POOL_STORE(act + off[q], v[q]);
      }
      pooled[row * row_items + j] = best;
      if (pool_idx != nullptr) pool_idx[row * row_items + j] = bi;
    }
  }
}

__global__ __launch_bounds__(kThreads) void maxpool_bwd_rows_kernel(const f32x4* __restrict__ d_pooled,
                                                                    const uint32_t* __restrict__ pool_idx, unsigned rows,
                                                                    unsigned Wo, unsigned CG, f32x4* __restrict__ d_act) {
  const unsigned row_items = Wo * CG;
  for (unsigned row = blockIdx.x; row < rows; row += gridDim.x) {
    const unsigned in0 = row * 4u * row_items;
    for (unsigned j = threadIdx.x; j < row_items; j += kThreads) {
      const unsigned cg = j % CG;
      const unsigned a = in0 + 2u * j - cg;
      const unsigned off[4] = {a, a + CG, a + 2u * row_items, a + 2u * row_items + CG};
      const uint32_t bi = pool_idx[row * row_items + j];
      const f32x4 d = d_pooled[row * row_items + j];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        bool any = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) any = any || ((bi >> (8 * k)) & 0xffu) == static_cast<uint32_t>(q);
        if (any) {  // untouched quads are neither read nor written
          f32x4 g = d_act[off[q]];
#pragma unroll
          for (int k = 0; k < 4; ++k) g[k] += (((bi >> (8 * k)) & 0xffu) == static_cast<uint32_t>(q)) ? d[k] : 0.f;
          d_act[off[q]] = g;
        }
      }
    }
  }
}

// ------------------------------------------------------------------ BatchNorm backward
// Grid stride is a multiple of CG, so a thread keeps one channel group for its whole loop.
template <int VEC>
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_kernel(const float* __restrict__ d_act,
                                                                 const float* __restrict__ y,
                                                                 const float* __restrict__ scale,
                                                                 const float* __restrict__ shift,
                                                                 const float* __restrict__ mean,
                                                                 const float* __restrict__ invstd, long items, int CG,
                                                                 float* __restrict__ partial) {
  using P = typename Pack<VEC>::T;
  __shared__ float sm[kThreads][VEC * 2];
  const long first = blockIdx.x * static_cast<long>(kThreads) + threadIdx.x;
  const int c = static_cast<int>(first % CG) * VEC;
  float sc[VEC], sh[VEC], mu[VEC], is[VEC], s1[VEC], s2[VEC];
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    sc[k] = scale[c + k];
    sh[k] = shift[c + k];
    mu[k] = mean[c + k];
    is[k] = invstd[c + k];
    s1[k] = 0.f;
    s2[k] = 0.f;
  }
  for (long i = first; i < items; i += static_cast<long>(gridDim.x) * kThreads) {
    const P g = reinterpret_cast<const P*>(d_act)[i];
    const P v = reinterpret_cast<const P*>(y)[i];
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const float yy = lane_of<VEC>(v, k);
      const float gg = (fmaf(yy, sc[k], sh[k]) > 0.f) ? lane_of<VEC>(g, k) : 0.f;
      s1[k] += gg;
      s2[k] += gg * (yy - mu[k]) * is[k];
    }
  }
#pragma unroll
  for (int k = 0; k < VEC; ++k) {
    sm[threadIdx.x][2 * k] = s1[k];
    sm[threadIdx.x][2 * k + 1] = s2[k];
  }
  __syncthreads();
  const int C = CG * VEC;
  const int base = static_cast<int>((blockIdx.x * static_cast<long>(kThreads)) % CG);
  for (int cc = threadIdx.x; cc < C; cc += kThreads) {
    const int cg = cc / VEC, k = cc % VEC;
    float a = 0.f, b = 0.f;
    for (int t = (cg - base + CG) % CG; t < kThreads; t += CG) {
      a += sm[t][2 * k];
      b += sm[t][2 * k + 1];
    }
    partial[(static_cast<long>(blockIdx.x) * C + cc) * 2 + 0] = a;
    partial[(static_cast<long>(blockIdx.x) * C + cc) * 2 + 1] = b;
  }
}

template <int VEC>
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(const float* __restrict__ d_act, const float* __restrict__ y,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift,
                                                                const float* __restrict__ mean,
                                                                const float* __restrict__ invstd,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ dgamma,
                                                                const float* __restrict__ dbeta, float inv_count,
                                                                long items, int CG, float* __restrict__ dy) {
  using P = typename Pack<VEC>::T;
  for (long i = blockIdx.x * static_cast<long>(kThreads) + threadIdx.x; i < items;
       i += static_cast<long>(gridDim.x) * kThreads) {
    const int c = static_cast<int>(i % CG) * VEC;
    const P g = reinterpret_cast<const P*>(d_act)[i];
    const P v = reinterpret_cast<const P*>(y)[i];
    P out;
#pragma unroll
    for (int k = 0; k < VEC; ++k) {
      const float yy = lane_of<VEC>(v, k);
      const float gg = (fmaf(yy, scale[c + k], shift[c + k]) > 0.f) ? lane_of<VEC>(g, k) : 0.f;
      const float xhat = (yy - mean[c + k]) * invstd[c + k];
      lane_of<VEC>(out, k) =
          gamma[c + k] * invstd[c + k] * (gg - dbeta[c + k] * inv_count - xhat * dgamma[c + k] * inv_count);
    }
    reinterpret_cast<P*>(dy)[i] = out;
  }
}

// ---- BatchNorm backward of an encoder node whose output was also max-pooled: the pooled consumer's gradient is routed
// to its argmax WHILE d_act is read (in both passes), which saves maxpool_bwd's read-modify-write pass over d_act.
// Row structured like the pool kernels: a workgroup walks image rows, a thread the (pixel, channel quad) items of a row;
// needs 4-aligned channels with a power-of-two quad count <= 256 (a thread keeps its quad), even H and W, tensors
// below 2^31 elements.  The global row index r = n * H + y has y's parity, and r >> 1 is the pooled row.
__device__ __forceinline__ f32x4 routed_grad(const f32x4* __restrict__ d_act, const f32x4* __restrict__ d_pooled,
                                             const uint32_t* __restrict__ pool_idx, unsigned row, unsigned x,
                                             unsigned cg, unsigned log2CG, unsigned row_items) {
  f32x4 g = d_act[row * row_items + (x << log2CG) + cg];
  const unsigned pitem = (row >> 1) * (row_items >> 1) + ((x >> 1) << log2CG) + cg;
  const uint32_t bi = pool_idx[pitem];
  const f32x4 dp = d_pooled[pitem];
  const uint32_t q = ((row & 1u) << 1) | (x & 1u);
#pragma unroll
  for (int k = 0; k < 4; ++k) g[k] += (((bi >> (8 * k)) & 0xffu) == q) ? dp[k] : 0.f;
  return g;
}

__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_pool_kernel(
    const f32x4* __restrict__ d_act, const f32x4* __restrict__ y, const f32x4* __restrict__ scale,
    const f32x4* __restrict__ shift, const f32x4* __restrict__ mean, const f32x4* __restrict__ invstd,
    const f32x4* __restrict__ d_pooled, const uint32_t* __restrict__ pool_idx, unsigned rows, unsigned W,
    unsigned log2CG, float* __restrict__ partial) {
  __shared__ float sm[kThreads][8];
  const unsigned CG = 1u << log2CG, cg = threadIdx.x & (CG - 1), row_items = W << log2CG;
  const f32x4 sc = scale[cg], sh = shift[cg], mu = mean[cg], is = invstd[cg];
  float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
  for (unsigned row = blockIdx.x; row < rows; row += gridDim.x) {
    for (unsigned j = threadIdx.x; j < row_items; j += kThreads) {
      const f32x4 g = routed_grad(d_act, d_pooled, pool_idx, row, j >> log2CG, cg, log2CG, row_items);
      const f32x4 v = y[row * row_items + j];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gg = (fmaf(v[k], sc[k], sh[k]) > 0.f) ? g[k] : 0.f;
        s1[k] += gg;
        s2[k] += gg * (v[k] - mu[k]) * is[k];
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    sm[threadIdx.x][2 * k] = s1[k];
    sm[threadIdx.x][2 * k + 1] = s2[k];
  }
  __syncthreads();
  const unsigned C = CG * 4;
  for (unsigned cc = threadIdx.x; cc < C; cc += kThreads) {
    const unsigned g4 = cc >> 2, k = cc & 3;
    float a = 0.f, b = 0.f;
    for (unsigned t = g4; t < kThreads; t += CG) {
      a += sm[t][2 * k];
      b += sm[t][2 * k + 1];
    }
    partial[(static_cast<long>(blockIdx.x) * C + cc) * 2 + 0] = a;
    partial[(static_cast<long>(blockIdx.x) * C + cc) * 2 + 1] = b;
  }
}

__global__ __launch_bounds__(kThreads) void bn_bwd_apply_pool_kernel(
    const f32x4* __restrict__ d_act, const f32x4* __restrict__ y, const f32x4* __restrict__ scale,
    const f32x4* __restrict__ shift, const f32x4* __restrict__ mean, const f32x4* __restrict__ invstd,
    const f32x4* __restrict__ gamma, const f32x4* __restrict__ dgamma, const f32x4* __restrict__ dbeta,
    const f32x4* __restrict__ d_pooled, const uint32_t* __restrict__ pool_idx, float inv_count, unsigned rows, unsigned W,
    unsigned log2CG, f32x4* __restrict__ dy) {
  const unsigned CG = 1u << log2CG, cg = threadIdx.x & (CG - 1), row_items = W << log2CG;
  const f32x4 sc = scale[cg], sh = shift[cg], mu = mean[cg], is = invstd[cg], ga = gamma[cg], dg = dgamma[cg],
              db = dbeta[cg];
  for (unsigned row = blockIdx.x; row < rows; row += gridDim.x) {
    for (unsigned j = threadIdx.x; j < row_items; j += kThreads) {
      const f32x4 g = routed_grad(d_act, d_pooled, pool_idx, row, j >> log2CG, cg, log2CG, row_items);
      const f32x4 v = y[row * row_items + j];
      f32x4 out;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const float gg = (fmaf(v[k], sc[k], sh[k]) > 0.f) ? g[k] : 0.f;
        const float xhat = (v[k] - mu[k]) * is[k];
        out[k] = ga[k] * is[k] * (gg - db[k] * inv_count - xhat * dg[k] * inv_count);
      }
      dy[row * row_items + j] = out;  // may alias d_act: the item was read by this thread above
    }
  }
}


__global__ __launch_bounds__(kThreads) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ weight,
                                                            const float* __restrict__ bias, long pixels, int HW, int C,
                                                            int n_cls, float keep_scale, uint32_t thr16, uint64_t seed,
                                                            const uint8_t* __restrict__ mask, const uint64_t* __restrict__ seed_dev, int use_drop,
                                                            float* __restrict__ out) {
  if (seed_dev != nullptr) seed += *seed_dev;  // graph-captured steps: the varying part of the seed lives in device memory
  __shared__ float wsm[kHeadMaxCls * kHeadMaxC];
  for (int i = threadIdx.x; i < n_cls * C; i += kThreads) wsm[i] = weight[i];
  __syncthreads();
  const int g4n = (C + 3) >> 2;
  for (long p = blockIdx.x * static_cast<long>(kThreads) + threadIdx.x; p < pixels;
       p += static_cast<long>(gridDim.x) * kThreads) {
    float acc[kHeadMaxCls];
#pragma unroll
    for (int k = 0; k < kHeadMaxCls; ++k) acc[k] = (k < n_cls) ? bias[k] : 0.f;
    const float* xp = x + p * C;
    for (int g = 0; g < g4n; ++g) {
      const uint64_t bits = (use_drop && mask == nullptr) ? keep_bits(seed, p, g4n, g) : 0;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = 4 * g + q;
        if (c < C) {
          float v = xp[c];
          if (use_drop) {
            const bool keep = (mask != nullptr) ? (mask[p * C + c] != 0) : keep_one(bits, q, thr16);
            v = keep ? v * keep_scale : 0.f;
          }
#pragma unroll
          for (int k = 0; k < kHeadMaxCls; ++k)
            if (k < n_cls) acc[k] += v * wsm[k * C + c];
        }
      }
    }
    const long n = p / HW, hw = p - n * HW;
#pragma unroll
    for (int k = 0; k < kHeadMaxCls; ++k)
      if (k < n_cls) out[(n * n_cls + k) * HW + hw] = 1.0f / (1.0f + expf(-acc[k]));
  }
}

// Forward head, coalesced: one wave per workgroup stages 64 pixels x C channels through LDS with full-line
// 16-byte loads (dropout applied on the way in), then lane = pixel reads its row (stride C+1: conflict-free)
// and the class weights come through scalar loads (uniform index).  Output is NCHW, coalesced along pixels.
__global__ __launch_bounds__(64) void head_fwd_tiled_kernel(const float* __restrict__ x, const float* __restrict__ weight,
                                                            const float* __restrict__ bias, long pixels, int HW, int C,
                                                            int n_cls, float keep_scale, uint32_t thr16, uint64_t seed,
                                                            const uint8_t* __restrict__ mask, const uint64_t* __restrict__ seed_dev, int use_drop,
                                                            float* __restrict__ out) {
  if (seed_dev != nullptr) seed += *seed_dev;  // graph-captured steps: the varying part of the seed lives in device memory
  extern __shared__ __attribute__((aligned(16))) float xs[];  // 64 * (C + 1) floats (sized by the launcher)
  const int lane = threadIdx.x;
  const int XS = C + 1, g4n = C >> 2;  // launcher guarantees C % 4 == 0
  const long n_tiles = (pixels + 63) / 64;
  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long p0 = tile * 64;
    __syncthreads();
    for (int it = lane; it < 64 * g4n; it += 64) {
      const int pl = it / g4n, gq = it - pl * g4n;
      const long p = p0 + pl;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (p < pixels) {
        v = *reinterpret_cast<const f32x4*>(x + p * C + 4 * gq);
        if (use_drop) {
          const uint64_t bits = (mask == nullptr) ? keep_bits(seed, p, g4n, gq) : 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const bool keep = (mask != nullptr) ? (mask[p * C + 4 * gq + q] != 0) : keep_one(bits, q, thr16);
            v[q] = keep ? v[q] * keep_scale : 0.f;
          }
        }
      }
      float* dst = &xs[pl * XS + 4 * gq];
      dst[0] = v[0];
      dst[1] = v[1];
      dst[2] = v[2];
      dst[3] = v[3];
    }
    __syncthreads();
    const long p = p0 + lane;
    float acc[kHeadMaxCls];
#pragma unroll
    for (int k = 0; k < kHeadMaxCls; ++k) acc[k] = (k < n_cls) ? bias[k] : 0.f;
    for (int c = 0; c < C; ++c) {
      const float v = xs[lane * XS + c];
#pragma unroll
      for (int k = 0; k < kHeadMaxCls; ++k)
        if (k < n_cls) acc[k] += v * weight[k * C + c];
    }
    if (p < pixels) {
      const long n = p / HW, hw = p - n * HW;
#pragma unroll
      for (int k = 0; k < kHeadMaxCls; ++k)
        if (k < n_cls) out[(n * n_cls + k) * HW + hw] = 1.0f / (1.0f + expf(-acc[k]));
    }
  }
}

// Forward head, streaming form for power-of-two channel-quad counts (C = 4 .. 128): lane = (pixel, channel quad), one
// coalesced 16-byte load per item, the class weights of the quad in registers.  The P per-class partial dot products
// of a lane are summed over the C/4 lanes of the pixel by a reduce-scatter (each exchange step halves the classes a
// lane still carries: P-1 + log2(G/P) shuffles instead of P log2 G), fixed order.  No LDS tile, no transposition.
// DROP: 0 = no dropout, 1 = keep flags from the counter hash, 2 = keep flags from a mask tensor -- three instantiations so
// that the loop body is straight-line code (as one kernel it carried ~16 uniform branches per item).  32-bit element
// offsets (the launcher takes this path for tensors below 2^31 elements); C = 4 G, so pixel -> element offset and
// pixel -> hash counter are shifts; the (image, position) pair of the NCHW output is carried along instead of divided
// out per item.
template <int LOG2G, int P, int DROP>  // P = classes padded to a power of two (4 or 8), P <= G
__global__ __launch_bounds__(kThreads) void head_fwd_stream_kernel(const float* __restrict__ x, const float* __restrict__ weight,
                                                                   const float* __restrict__ bias, unsigned pixels, unsigned HW,
                                                                   int n_cls, float keep_scale, uint32_t thr16, uint64_t seed,
                                                                   const uint8_t* __restrict__ mask, const uint64_t* __restrict__ seed_dev, float* __restrict__ out) {
  if (seed_dev != nullptr) seed += *seed_dev;  // graph-captured steps: the varying part of the seed lives in device memory
  constexpr int G = 1 << LOG2G;  // lanes (channel quads) per pixel
  const int gq = threadIdx.x & (G - 1);
  f32x4 wq[P];
#pragma unroll
  for (int k = 0; k < P; ++k)
    wq[k] = (k < n_cls) ? *reinterpret_cast<const f32x4*>(weight + k * 4 * G + 4 * gq) : f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr unsigned ppb = kThreads >> LOG2G;  // pixels per block and slot
  constexpr int U = 4;                         // pixels per thread and iteration: 4 loads in flight
  const unsigned span = gridDim.x * ppb, outer = U * span;
  const unsigned pl = threadIdx.x >> LOG2G;
  // (image, position) of this thread's first pixel and the step of one `span`, kept up to date by adds
  const unsigned span_n = span / HW, span_hw = span - span_n * HW;
  unsigned p0 = blockIdx.x * ppb + pl;
  unsigned n0 = p0 / HW, hw0 = p0 - n0 * HW;
  for (; p0 - pl < pixels; p0 += outer) {  // wave-uniform trip count
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned p = p0 + u * span;
      v[u] = (p < pixels) ? *reinterpret_cast<const f32x4*>(x + ((p << (LOG2G + 2)) + 4 * gq)) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const unsigned p = p0 + u * span;
      const bool valid = p < pixels;
      if constexpr (DROP == 1) {
        const uint64_t bits = keep_bits(seed, p, G, gq);
#pragma unroll
        for (int q = 0; q < 4; ++q) v[u][q] = keep_one(bits, q, thr16) ? v[u][q] * keep_scale : 0.f;
      } else if constexpr (DROP == 2) {
        const uint32_t m4 = valid ? *reinterpret_cast<const uint32_t*>(mask + ((p << (LOG2G + 2)) + 4 * gq)) : 0u;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[u][q] = ((m4 >> (8 * q)) & 0xffu) != 0 ? v[u][q] * keep_scale : 0.f;
      }
      float acc[P];
#pragma unroll
      for (int k = 0; k < P; ++k)
        acc[k] = fmaf(v[u][0], wq[k][0], fmaf(v[u][1], wq[k][1], fmaf(v[u][2], wq[k][2], v[u][3] * wq[k][3])));
      // reduce-scatter over the G lanes of the pixel: with `live` classes left, a lane keeps the half selected by its
      // bit `off` and adds the partner's partials of that half; once one class is left, a plain butterfly sum
      int cls = 0;  // class this lane ends up with
      static_for<LOG2G>([&](auto sc) {
        constexpr int step = decltype(sc)::v, off = G >> (1 + step);
        constexpr int live = (P >> step) > 1 ? (P >> step) : 1;  // classes a lane still carries before this step
        if constexpr (live > 1) {
          constexpr int half = live >> 1;
          const bool upper = (gq & off) != 0;
#pragma unroll
          for (int i = 0; i < half; ++i) {
            const float send = upper ? acc[i] : acc[half + i];
            const float keep = upper ? acc[half + i] : acc[i];
            acc[i] = keep + xor_lane<off>(send);
          }
          cls += upper ? half : 0;
        } else {
          acc[0] += xor_lane<off>(acc[0]);
        }
      });
      // this item's (image, position): u steps of `span` from the thread's first pixel of the iteration
      unsigned n = n0 + u * span_n, hw = hw0 + u * span_hw;
#pragma unroll
      for (int c = 0; c < U - 1; ++c) {  // at most u carries
        const bool carry = c < u && hw >= HW;
        hw -= carry ? HW : 0u;
        n += carry ? 1u : 0u;
      }
      constexpr int kDup = (G > P) ? G / P : 1;  // lanes that end with the same class total
      if (valid && cls < n_cls && (gq & (kDup - 1)) == 0)
        out[(static_cast<long>(n) * n_cls + cls) * HW + hw] = 1.0f / (1.0f + __expf(-(acc[0] + bias[cls])));
    }
    // advance the carried position by U spans
    n0 += U * span_n;
    hw0 += U * span_hw;
#pragma unroll
    for (int c = 0; c < U; ++c) {
      const bool carry = hw0 >= HW;
      hw0 -= carry ? HW : 0u;
      n0 += carry ? 1u : 0u;
    }
  }
}

// tile = 64 consecutive pixels.  LDS: x*keep*scale [64][C+1], dlogit [64][8], W [8][C].
__global__ __launch_bounds__(kThreads) void head_bwd_kernel(const float* __restrict__ d_out, const float* __restrict__ outp,
                                                            const float* __restrict__ x, const float* __restrict__ weight,
                                                            long pixels, int HW, int C, int n_cls, float keep_scale,
                                                            uint32_t thr16, uint64_t seed, const uint8_t* __restrict__ mask, const uint64_t* __restrict__ seed_dev,
                                                            int use_drop, float* __restrict__ dx, int accumulate, int gate_x,
                                                            float* __restrict__ partial) {
  if (seed_dev != nullptr) seed += *seed_dev;  // graph-captured steps: the varying part of the seed lives in device memory
  __shared__ float xs[64 * (kHeadMaxC + 1)];
  __shared__ float dl[64 * kHeadMaxCls];
  __shared__ float wsm[kHeadMaxCls * kHeadMaxC];
  const int tid = threadIdx.x;
  const int XS = C + 1;
  const int g4n = (C + 3) >> 2;
  for (int i = tid; i < n_cls * C; i += kThreads) wsm[i] = weight[i];
  float wacc[4] = {0.f, 0.f, 0.f, 0.f};  // dW entries tid, tid+256, ... (n_cls*C <= 1024)
  float bacc = 0.f;                      // db entry tid (< n_cls)
  const long n_tiles = (pixels + 63) / 64;
  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long p0 = tile * 64;
    __syncthreads();
    // dlogit = d_out * out * (1 - out)
    for (int it = tid; it < 64 * n_cls; it += kThreads) {
      const int pl = it & 63, k = it >> 6;
      const long p = p0 + pl;
      float v = 0.f;
      if (p < pixels) {
        const long n = p / HW, hw = p - n * HW;
        const long o = (n * n_cls + k) * HW + hw;
        const float pr = outp[o];
        v = d_out[o] * pr * (1.f - pr);
      }
      dl[pl * kHeadMaxCls + k] = v;
    }
    // x * keep * scale
    for (int it = tid; it < 64 * C; it += kThreads) {
      const int pl = it / C, c = it - pl * C;
      const long p = p0 + pl;
      float v = 0.f;
      if (p < pixels) {
        v = x[p * C + c];
        if (use_drop) {
          const bool keep = (mask != nullptr) ? (mask[p * C + c] != 0)
                                              : keep_one(keep_bits(seed, p, g4n, c >> 2), c & 3, thr16);
          v = keep ? v * keep_scale : 0.f;
        }
      }
      xs[pl * XS + c] = v;
    }
    __syncthreads();
    // dx[p, c] = keep * scale * sum_k W[k, c] * dlogit[p, k]
    for (int it = tid; it < 64 * C; it += kThreads) {
      const int pl = it / C, c = it - pl * C;
      const long p = p0 + pl;
      if (p < pixels) {
        float s = 0.f;
#pragma unroll
        for (int k = 0; k < kHeadMaxCls; ++k)
          if (k < n_cls) s += wsm[k * C + c] * dl[pl * kHeadMaxCls + k];
        if (use_drop) {
          const bool keep = (mask != nullptr) ? (mask[p * C + c] != 0)
                                              : keep_one(keep_bits(seed, p, g4n, c >> 2), c & 3, thr16);
          s = keep ? s * keep_scale : 0.f;
        }
        if (accumulate) s += dx[p * C + c];
        if (gate_x) s = (x[p * C + c] > 0.f) ? s : 0.f;
        dx[p * C + c] = s;
      }
    }
    // dW[k, c] += sum_p dlogit[p, k] * xs[p, c];  db[k] += sum_p dlogit[p, k]
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int idx = tid + q * kThreads;
      if (idx < n_cls * C) {
        const int k = idx / C, c = idx - k * C;
        float s = 0.f;
        for (int pl = 0; pl < 64; ++pl) s += dl[pl * kHeadMaxCls + k] * xs[pl * XS + c];
        wacc[q] += s;
      }
    }
    if (tid < n_cls) {
      float s = 0.f;
      for (int pl = 0; pl < 64; ++pl) s += dl[pl * kHeadMaxCls + tid];
      bacc += s;
    }
  }
  float* dst = partial + static_cast<long>(blockIdx.x) * (n_cls * C + n_cls);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int idx = tid + q * kThreads;
    if (idx < n_cls * C) dst[idx] = wacc[q];
  }
  if (tid < n_cls) dst[n_cls * C + tid] = bacc;
}

// Backward head, vectorised (C % 4 == 0): per 64-pixel tile
//   1. dlogit = d_out * out * (1 - out)                    (NCHW reads, coalesced along pixels) -> LDS
//   2. one pass over x in 16-byte pieces: dropout keep mask from ONE hash per piece, x*keep*scale -> LDS for
//      the weight gradient, and dx = keep*scale * (W^T dlogit) (+ old dx, ReLU gate) written straight back
//   3. dW[k, c] += sum_p dlogit[p, k] * xs[p, c]: all 256 threads, two pixel halves per (k, c)
// Dynamic LDS: xs [64][C+1] | dlogit [64][8] | W [8][C] | scratch [2][n_cls*C].
// (launch bound of 4 waves per SIMD: left alone hipcc unrolls the reduction loops into 256 VGPRs and the kernel
// runs at 2 workgroups per CU, latency-bound at 1 TB/s)
__global__ __launch_bounds__(kThreads, 4) void head_bwd_vec_kernel(const float* __restrict__ d_out,
                                                                const float* __restrict__ outp,
                                                                const float* __restrict__ x,
                                                                const float* __restrict__ weight, long pixels, int HW,
                                                                int C, int n_cls, float keep_scale, uint32_t thr16,
                                                                uint64_t seed, const uint8_t* __restrict__ mask, const uint64_t* __restrict__ seed_dev,
                                                                int use_drop, float* __restrict__ dx, int accumulate,
                                                                int gate_x, float* __restrict__ partial) {
  if (seed_dev != nullptr) seed += *seed_dev;  // graph-captured steps: the varying part of the seed lives in device memory
  extern __shared__ __attribute__((aligned(16))) float hsm[];
  const int XS = C + 1, g4n = C >> 2, NW = n_cls * C;
  float* xs = hsm;
  float* dl = xs + 64 * XS;
  float* wsm = dl + 64 * kHeadMaxCls;
  float* scratch = wsm + kHeadMaxCls * C;
  const int tid = threadIdx.x;
  for (int i = tid; i < NW; i += kThreads) wsm[i] = weight[i];
  constexpr int kMaxPairs = (kHeadMaxCls * kHeadMaxC + 127) / 128;  // (k, c) pairs per thread
  float wacc[kMaxPairs];
#pragma unroll
  for (int q = 0; q < kMaxPairs; ++q) wacc[q] = 0.f;
  float bacc = 0.f;
  const int pair0 = tid & 127, half = tid >> 7;
  const long n_tiles = (pixels + 63) / 64;
  for (long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long p0 = tile * 64;
    __syncthreads();
    for (int it = tid; it < 64 * n_cls; it += kThreads) {
      const int pl = it & 63, k = it >> 6;
      const long p = p0 + pl;
      float v = 0.f;
      if (p < pixels) {
        const long n = p / HW, hw = p - n * HW;
        const long o = (n * n_cls + k) * HW + hw;
        const float pr = outp[o];
        v = d_out[o] * pr * (1.f - pr);
      }
      dl[pl * kHeadMaxCls + k] = v;
    }
    __syncthreads();
    for (int it = tid; it < 64 * g4n; it += kThreads) {
      const int pl = it / g4n, gq = it - pl * g4n;
      const long p = p0 + pl;
      f32x4 xv = {0.f, 0.f, 0.f, 0.f};
      float ms[4] = {1.f, 1.f, 1.f, 1.f};
      if (p < pixels) {
        xv = *reinterpret_cast<const f32x4*>(x + p * C + 4 * gq);
        if (use_drop) {
          const uint64_t bits = (mask == nullptr) ? keep_bits(seed, p, g4n, gq) : 0;
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const bool keep = (mask != nullptr) ? (mask[p * C + 4 * gq + q] != 0) : keep_one(bits, q, thr16);
            ms[q] = keep ? keep_scale : 0.f;
          }
        }
        f32x4 s = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < kHeadMaxCls; ++k) {
          if (k < n_cls) {
            const float dk = dl[pl * kHeadMaxCls + k];
#pragma unroll
            for (int q = 0; q < 4; ++q) s[q] += wsm[k * C + 4 * gq + q] * dk;
          }
        }
        float* dst = dx + p * C + 4 * gq;
        f32x4 old = {0.f, 0.f, 0.f, 0.f};
        if (accumulate) old = *reinterpret_cast<const f32x4*>(dst);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v = s[q] * ms[q] + old[q];
          if (gate_x) v = (xv[q] > 0.f) ? v : 0.f;
          s[q] = v;
        }
        *reinterpret_cast<f32x4*>(dst) = s;
      }
      float* xd = &xs[pl * XS + 4 * gq];
#pragma unroll
      for (int q = 0; q < 4; ++q) xd[q] = xv[q] * ms[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kMaxPairs; ++q) {
      const int idx = pair0 + q * 128;
      if (idx < NW) {
        const int k = idx / C, c = idx - k * C;
        float s = 0.f;
#pragma unroll 4
        for (int pl = 32 * half; pl < 32 * half + 32; ++pl) s += dl[pl * kHeadMaxCls + k] * xs[pl * XS + c];
        wacc[q] += s;
      }
    }
    if (tid < n_cls) {
      float s = 0.f;
      for (int pl = 0; pl < 64; ++pl) s += dl[pl * kHeadMaxCls + tid];
      bacc += s;
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kMaxPairs; ++q) {
    const int idx = pair0 + q * 128;
    if (idx < NW) scratch[half * NW + idx] = wacc[q];
  }
  __syncthreads();
  float* dst = partial + static_cast<long>(blockIdx.x) * (NW + n_cls);
  for (int i = tid; i < NW; i += kThreads) dst[i] = scratch[i] + scratch[NW + i];
  if (tid < n_cls) dst[NW + tid] = bacc;
}

// The same for C = 4 * 2^LOG2G channels and tensors below 2^31 elements (every configuration of the reference): the
// index arithmetic is shifts and 32-bit, a thread's channel quad is the same for all its pieces so its class weights
// live in registers (the general kernel re-reads them from LDS per piece: 16 + 4 LDS reads per 16 bytes of x), the
// (class, channel) pairs of the weight-gradient pass are decoded once instead of once per tile, and the dropout mode is
// a template parameter (0 none, 1 counter hash, 2 mask tensor) so that the piece loop is straight-line code.
template <int LOG2G, int DROP, int PCLS>  // PCLS = classes padded to 4 or 8 (zero weights past n_cls: no class branches)
__global__ __launch_bounds__(kThreads, 4) void head_bwd_pow2_kernel(const float* __restrict__ d_out,
                                                                 const float* __restrict__ outp,
                                                                 const float* __restrict__ x,
                                                                 const float* __restrict__ weight, unsigned pixels,
                                                                 unsigned HW, int n_cls, float keep_scale, uint32_t thr16,
                                                                 uint64_t seed, const uint8_t* __restrict__ mask, const uint64_t* __restrict__ seed_dev,
                                                                 float* __restrict__ dx, int accumulate, int gate_x,
                                                                 float* __restrict__ partial) {
  if (seed_dev != nullptr) seed += *seed_dev;  // graph-captured steps: the varying part of the seed lives in device memory
  extern __shared__ __attribute__((aligned(16))) float hsm[];
  constexpr int G = 1 << LOG2G, C = 4 * G, XS = C + 1;
  constexpr int ITEMS = (64 * G + kThreads - 1) / kThreads;  // 16-byte pieces of a 64-pixel tile per thread
  const int NW = n_cls * C;
  float* xs = hsm;
  float* dl = xs + 64 * XS;
  float* scratch = dl + 64 * kHeadMaxCls;
  const int tid = threadIdx.x;
  const int gq = tid & (G - 1);  // channel quad of every piece of this thread (kThreads is a multiple of G)
  f32x4 wq[PCLS];
#pragma unroll
  for (int k = 0; k < PCLS; ++k)
    wq[k] = (k < n_cls) ? *reinterpret_cast<const f32x4*>(weight + k * C + 4 * gq) : f32x4{0.f, 0.f, 0.f, 0.f};
  constexpr int kMaxPairs = (PCLS * C + 127) / 128;  // (k, c) pairs per thread
  float wacc[kMaxPairs];
  int pair_k[kMaxPairs], pair_c[kMaxPairs];
  const int pair0 = tid & 127, half = tid >> 7;
#pragma unroll
  for (int q = 0; q < kMaxPairs; ++q) {
    wacc[q] = 0.f;
    const int idx = pair0 + q * 128;
    pair_k[q] = idx >> (LOG2G + 2);
    pair_c[q] = idx & (C - 1);
  }
  float bacc = 0.f;
  for (int i = tid; i < 64 * kHeadMaxCls; i += kThreads) dl[i] = 0.f;  // classes past n_cls are never written again
  const unsigned n_tiles = (pixels + 63) / 64;
  for (unsigned tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const unsigned p0 = tile * 64;
    __syncthreads();
    for (int it = tid; it < 64 * n_cls; it += kThreads) {
      const int pl = it & 63, k = it >> 6;
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
      if (ITEMS * kThreads != 64 * G && it >= 64 * G) break;
      const int pl = it >> LOG2G;
      const unsigned p = p0 + pl;
      f32x4 xv = {0.f, 0.f, 0.f, 0.f};
      float ms[4] = {1.f, 1.f, 1.f, 1.f};
      if (p < pixels) {
        const unsigned off = (p << (LOG2G + 2)) + 4 * gq;
        xv = *reinterpret_cast<const f32x4*>(x + off);
        if constexpr (DROP == 1) {
          const uint64_t bits = keep_bits(seed, p, G, gq);
#pragma unroll
          for (int q = 0; q < 4; ++q) ms[q] = keep_one(bits, q, thr16) ? keep_scale : 0.f;
        } else if constexpr (DROP == 2) {
          const uint32_t m4 = *reinterpret_cast<const uint32_t*>(mask + off);
#pragma unroll
          for (int q = 0; q < 4; ++q) ms[q] = ((m4 >> (8 * q)) & 0xffu) != 0 ? keep_scale : 0.f;
        }
        f32x4 sacc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < PCLS; ++k) {
          const float dk = dl[pl * kHeadMaxCls + k];
#pragma unroll
          for (int q = 0; q < 4; ++q) sacc[q] += wq[k][q] * dk;
        }
        f32x4 old = {0.f, 0.f, 0.f, 0.f};
        if (accumulate) old = *reinterpret_cast<const f32x4*>(dx + off);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v = sacc[q] * ms[q] + old[q];
          if (gate_x) v = (xv[q] > 0.f) ? v : 0.f;
          sacc[q] = v;
        }
        *reinterpret_cast<f32x4*>(dx + off) = sacc;
      }
      float* xd = &xs[pl * XS + 4 * gq];
#pragma unroll
      for (int q = 0; q < 4; ++q) xd[q] = xv[q] * ms[q];
    }
    __syncthreads();
#pragma unroll
    for (int q = 0; q < kMaxPairs; ++q) {
      if (pair0 + q * 128 < NW) {
        const int k = pair_k[q], c = pair_c[q];
        float sum = 0.f;
#pragma unroll 4
        for (int pl = 32 * half; pl < 32 * half + 32; ++pl) sum += dl[pl * kHeadMaxCls + k] * xs[pl * XS + c];
        wacc[q] += sum;
      }
    }
    if (tid < n_cls) {
      float sum = 0.f;
      for (int pl = 0; pl < 64; ++pl) sum += dl[pl * kHeadMaxCls + tid];
      bacc += sum;
    }
  }
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kMaxPairs; ++q) {
    const int idx = pair0 + q * 128;
    if (idx < NW) scratch[half * NW + idx] = wacc[q];
  }
  __syncthreads();
  float* dst = partial + static_cast<long>(blockIdx.x) * (NW + n_cls);
  for (int i = tid; i < NW; i += kThreads) dst[i] = scratch[i] + scratch[NW + i];
  if (tid < n_cls) dst[NW + tid] = bacc;
}

__global__ void sum_partials_kernel(const float* __restrict__ partial, long n_blocks, long len, float* __restrict__ out) {
  // 16 outputs x 64 row groups per workgroup (1024 threads): a thread adds n_blocks / 64 rows (4 loads in flight);
  // the groups are combined through LDS in fixed order.  fp64 sums: thousands of same-sign partials.
  __shared__ double part[64][16];
  const int e = threadIdx.x & 15, g = threadIdx.x >> 4;
  const long i = blockIdx.x * 16L + e;
  double s = 0.0;
  if (i < len) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    long b = g;
    for (; b + 192 < n_blocks; b += 256) {
      s0 += static_cast<double>(partial[b * len + i]);
      s1 += static_cast<double>(partial[(b + 64) * len + i]);
      s2 += static_cast<double>(partial[(b + 128) * len + i]);
      s3 += static_cast<double>(partial[(b + 192) * len + i]);
    }
    for (; b < n_blocks; b += 64) s0 += static_cast<double>(partial[b * len + i]);
    s = (s0 + s1) + (s2 + s3);
  }
  part[g][e] = s;
  __syncthreads();
  if (g == 0 && i < len) {
    double t = 0.0;
#pragma unroll
    for (int k = 0; k < 64; ++k) t += part[k][e];
    out[i] = static_cast<float>(t);
  }
}

// ------------------------------------------------------------------ bilinear x2, align_corners = True
__device__ __forceinline__ void bilinear_src(int dst, int in_size, float rscale, int& i0, int& i1, float& l1) {
  const float s = rscale * static_cast<float>(dst);
  i0 = static_cast<int>(s);
  if (i0 > in_size - 1) i0 = in_size - 1;
  i1 = i0 + ((i0 < in_size - 1) ? 1 : 0);
  l1 = s - static_cast<float>(i0);
}

__global__ __launch_bounds__(kThreads) void bilinear2x_fwd_kernel(const float* __restrict__ x, int N, int H, int W, int C,
                                                                  float* __restrict__ y) {
  const int Ho = 2 * H, Wo = 2 * W;
  const float ry = (Ho > 1) ? static_cast<float>(H - 1) / static_cast<float>(Ho - 1) : 0.f;
  const float rx = (Wo > 1) ? static_cast<float>(W - 1) / static_cast<float>(Wo - 1) : 0.f;
  const long items = static_cast<long>(N) * Ho * Wo * C;
  for (long i = blockIdx.x * static_cast<long>(kThreads) + threadIdx.x; i < items;
       i += static_cast<long>(gridDim.x) * kThreads) {
    const int c = static_cast<int>(i % C);
    long r = i / C;
    const int xo = static_cast<int>(r % Wo);
    r /= Wo;
    const int yo = static_cast<int>(r % Ho);
    const int n = static_cast<int>(r / Ho);
    int y0, y1, x0, x1;
    float ly, lx;
    bilinear_src(yo, H, ry, y0, y1, ly);
    bilinear_src(xo, W, rx, x0, x1, lx);
    const float* b = x + static_cast<long>(n) * H * W * C + c;
    const float v00 = b[(static_cast<long>(y0) * W + x0) * C], v01 = b[(static_cast<long>(y0) * W + x1) * C];
    const float v10 = b[(static_cast<long>(y1) * W + x0) * C], v11 = b[(static_cast<long>(y1) * W + x1) * C];
    y[i] = (1.f - ly) * ((1.f - lx) * v00 + lx * v01) + ly * ((1.f - lx) * v10 + lx * v11);
  }
}

// gather form of the transposed stencil: every source pixel visits the <= 5x5 destination pixels that can
// reference it and recomputes their weights, so the sum has a fixed order (no atomics).
__global__ __launch_bounds__(kThreads) void bilinear2x_bwd_kernel(const float* __restrict__ dy, int N, int H, int W, int C,
                                                                  float* __restrict__ dx, int accumulate) {
  const int Ho = 2 * H, Wo = 2 * W;
  const float ry = (Ho > 1) ? static_cast<float>(H - 1) / static_cast<float>(Ho - 1) : 0.f;
  const float rx = (Wo > 1) ? static_cast<float>(W - 1) / static_cast<float>(Wo - 1) : 0.f;
  const long items = static_cast<long>(N) * H * W * C;
  for (long i = blockIdx.x * static_cast<long>(kThreads) + threadIdx.x; i < items;
       i += static_cast<long>(gridDim.x) * kThreads) {
    const int c = static_cast<int>(i % C);
    long r = i / C;
    const int xs = static_cast<int>(r % W);
    r /= W;
    const int ys = static_cast<int>(r % H);
    const int n = static_cast<int>(r / H);
    const int ylo = max(0, 2 * ys - 3), yhi = min(Ho - 1, 2 * ys + 3);
    const int xlo = max(0, 2 * xs - 3), xhi = min(Wo - 1, 2 * xs + 3);
    float s = 0.f;
    for (int yo = ylo; yo <= yhi; ++yo) {
      int y0, y1;
      float ly;
      bilinear_src(yo, H, ry, y0, y1, ly);
      float wy = 0.f;
      if (y0 == ys) wy += 1.f - ly;
      if (y1 == ys) wy += ly;
      if (wy == 0.f) continue;
      for (int xo = xlo; xo <= xhi; ++xo) {
        int x0, x1;
        float lx;
        bilinear_src(xo, W, rx, x0, x1, lx);
        float wx = 0.f;
        if (x0 == xs) wx += 1.f - lx;
        if (x1 == xs) wx += lx;
        if (wx != 0.f) s += wy * wx * dy[((static_cast<long>(n) * Ho + yo) * Wo + xo) * C + c];
      }
    }
    dx[i] = accumulate ? dx[i] + s : s;
  }
}

// ------------------------------------------------------------------ layout converters
__global__ __launch_bounds__(kThreads) void nchw_to_nhwc_kernel(const float* __restrict__ src, int N, int C, long HW,
                                                                float* __restrict__ dst) {
  const long items = static_cast<long>(N) * C * HW;
  for (long i = blockIdx.x * static_cast<long>(kThreads) + threadIdx.x; i < items;
       i += static_cast<long>(gridDim.x) * kThreads) {
    const int c = static_cast<int>(i % C);
    const long r = i / C;
    const long hw = r % HW, n = r / HW;
    dst[i] = src[(n * C + c) * HW + hw];
  }
}
__global__ __launch_bounds__(kThreads) void nhwc_to_nchw_kernel(const float* __restrict__ src, int N, int C, long HW,
                                                                float* __restrict__ dst) {
  const long items = static_cast<long>(N) * C * HW;
  for (long i = blockIdx.x * static_cast<long>(kThreads) + threadIdx.x; i < items;
       i += static_cast<long>(gridDim.x) * kThreads) {
    const long hw = i % HW;
    const long r = i / HW;
    const int c = static_cast<int>(r % C);
    const long n = r / C;
    dst[i] = src[(n * HW + hw) * C + c];
  }
}

inline unsigned fin_threads(long n_blocks) {  // workgroup size of the per-channel reductions over partial rows
  return n_blocks >= 4096 ? 1024u : n_blocks >= 1024 ? 512u : 256u;
}
inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace
}  // namespace unetpp

using namespace unetpp;
#define ST(s) static_cast<hipStream_t>(s)

extern "C" int unetpp_bn_finalize(const float* partial, int64_t n_blocks, int32_t C, int64_t count, const float* gamma,
                                  const float* beta, float eps, float momentum, float* running_mean,
                                  float* running_var, float* mean, float* invstd, float* scale, float* shift,
                                  void* stream) {
  if (!partial || n_blocks < 1 || C < 1 || count < 1 || !gamma || !beta || !mean || !invstd || !scale || !shift)
    return UNETPP_EINVAL;
  if ((running_mean == nullptr) != (running_var == nullptr)) return UNETPP_EINVAL;
  if ((reinterpret_cast<uintptr_t>(partial) & 7) != 0) return UNETPP_EINVAL;  // rows are read as (sum, sum of squares) pairs
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(fin_threads(n_blocks)), 0, ST(stream), partial, n_blocks, C,
                     count, gamma, beta, eps, momentum, running_mean, running_var, mean, invstd, scale, shift);
  return launch_status();
}

extern "C" int unetpp_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                     const float* running_var, float eps, int32_t C, float* scale, float* shift,
                                     void* stream) {
  if (!gamma || !beta || !running_mean || !running_var || C < 1 || !scale || !shift) return UNETPP_EINVAL;
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((C + 63) / 64), dim3(64), 0, ST(stream), gamma, beta, running_mean,
                     running_var, eps, C, scale, shift);
  return launch_status();
}

namespace {
// the row-structured pool kernels: 32-bit element offsets, rows long enough to occupy a workgroup
inline bool rows_form_ok(int N, int H, int W, int C) {
  return static_cast<long>(N) * H * W * C < 0x7fffffffL && static_cast<long>(W / 2) * (C / 4) >= 64;
}
}  // namespace

extern "C" int unetpp_affine_relu_pool(const float* y, const float* scale, const float* shift, int32_t relu, int32_t N,
                                       int32_t H, int32_t W, int32_t C, float* act, float* pooled, uint8_t* pool_idx,
                                       void* stream) {
  if (!y || N < 1 || H < 1 || W < 1 || C < 1) return UNETPP_EINVAL;
  if ((scale == nullptr) != (shift == nullptr)) return UNETPP_EINVAL;
  if (act == nullptr && pooled == nullptr) return UNETPP_EINVAL;
  const bool vec = (C % 4 == 0) && aligned16(y) && (!act || aligned16(act)) && (!pooled || aligned16(pooled));
  if (pooled == nullptr) {
    const long pixels = static_cast<long>(N) * H * W;
    if (vec) {
      const long items = pixels * (C / 4);
      hipLaunchKernelGGL(affine_relu_kernel<4>, dim3(grid_for(items)), dim3(kThreads), 0, ST(stream), y, scale, shift,
                         relu, items, C / 4, act);
    } else {
      const long items = pixels * C;
      hipLaunchKernelGGL(affine_relu_kernel<1>, dim3(grid_for(items)), dim3(kThreads), 0, ST(stream), y, scale, shift,
                         relu, items, C, act);
    }
    return launch_status();
  }
  if (H < 2 || W < 2) return UNETPP_EINVAL;  // (pool_idx may be NULL: forward-only callers need no winners)
  if ((H & 1) || (W & 1)) {
    // nn.MaxPool2d(2) floors (the classic UNet on sizes that are not multiples of 16, models/unet.py:40-46): the last
    // odd row / column is in no window, but the activation is wanted for ALL pixels -- apply pass over the whole
    // tensor first, then the window kernel on the activation (its index arithmetic uses H, W as strides and H/2, W/2 as
    // the window grid; the winners are those of the transformed values either way)
    const float* src = y;
    if (act != nullptr) {
      const int rc = unetpp_affine_relu_pool(y, scale, shift, relu, N, H, W, C, act, nullptr, nullptr, stream);
      if (rc != UNETPP_OK) return rc;
      src = act;
      scale = shift = nullptr;
      relu = 0;
    }
    const long win = static_cast<long>(N) * (H / 2) * (W / 2);
    const bool v4 = (C % 4 == 0) && aligned16(src) && aligned16(pooled);
    if (v4)
      hipLaunchKernelGGL(affine_relu_pool_kernel<4>, dim3(grid_for(win * (C / 4))), dim3(kThreads), 0, ST(stream), src, scale,
                         shift, relu, N, H, W, C / 4, static_cast<float*>(nullptr), pooled, pool_idx);
    else
      hipLaunchKernelGGL(affine_relu_pool_kernel<1>, dim3(grid_for(win * C)), dim3(kThreads), 0, ST(stream), src, scale, shift,
                         relu, N, H, W, C, static_cast<float*>(nullptr), pooled, pool_idx);
    return launch_status();
  }
  const long windows = static_cast<long>(N) * (H / 2) * (W / 2);
  const unsigned rows = static_cast<unsigned>(N * (H / 2)), row_items = static_cast<unsigned>((W / 2) * (C / 4));
  if (vec && rows_form_ok(N, H, W, C) && (scale == nullptr || (aligned16(scale) && aligned16(shift))) &&
      (reinterpret_cast<uintptr_t>(pool_idx) & 3) == 0)
    hipLaunchKernelGGL(affine_relu_pool_rows_kernel, dim3(rows < 16384u ? rows : 16384u), dim3(kThreads), 0, ST(stream),
                       reinterpret_cast<const f32x4*>(y), reinterpret_cast<const f32x4*>(scale),
                       reinterpret_cast<const f32x4*>(shift), relu, rows, static_cast<unsigned>(W / 2),
                       static_cast<unsigned>(C / 4), reinterpret_cast<f32x4*>(act), reinterpret_cast<f32x4*>(pooled),
                       reinterpret_cast<uint32_t*>(pool_idx));
  else if (vec)
    hipLaunchKernelGGL(affine_relu_pool_kernel<4>, dim3(grid_for(windows * (C / 4))), dim3(kThreads), 0, ST(stream), y,
                       scale, shift, relu, N, H, W, C / 4, act, pooled, pool_idx);
  else
    hipLaunchKernelGGL(affine_relu_pool_kernel<1>, dim3(grid_for(windows * C)), dim3(kThreads), 0, ST(stream), y, scale,
                       shift, relu, N, H, W, C, act, pooled, pool_idx);
  return launch_status();
}

extern "C" int unetpp_maxpool_bwd(const float* d_pooled, const uint8_t* pool_idx, int32_t N, int32_t H, int32_t W,
                                  int32_t C, float* d_act, void* stream) {
  if (!d_pooled || !pool_idx || !d_act || N < 1 || H < 2 || W < 2 || C < 1) return UNETPP_EINVAL;
  const long windows = static_cast<long>(N) * (H / 2) * (W / 2);  // odd H / W: the last row / column is in no window (floor)
  const unsigned rows = static_cast<unsigned>(N * (H / 2));
  if (!(H & 1) && !(W & 1) && C % 4 == 0 && rows_form_ok(N, H, W, C) && aligned16(d_pooled) && aligned16(d_act) &&
      (reinterpret_cast<uintptr_t>(pool_idx) & 3) == 0)
    hipLaunchKernelGGL(maxpool_bwd_rows_kernel, dim3(rows < 16384u ? rows : 16384u), dim3(kThreads), 0, ST(stream),
                       reinterpret_cast<const f32x4*>(d_pooled), reinterpret_cast<const uint32_t*>(pool_idx), rows,
                       static_cast<unsigned>(W / 2), static_cast<unsigned>(C / 4), reinterpret_cast<f32x4*>(d_act));
  else if (C % 4 == 0)
    hipLaunchKernelGGL(maxpool_bwd_kernel<4>, dim3(grid_for(windows * (C / 4))), dim3(kThreads), 0, ST(stream), d_pooled,
                       pool_idx, N, H, W, C / 4, d_act);
  else
    hipLaunchKernelGGL(maxpool_bwd_kernel<1>, dim3(grid_for(windows * C)), dim3(kThreads), 0, ST(stream), d_pooled,
                       pool_idx, N, H, W, C, d_act);
  return launch_status();
}

namespace {
// blocks for the BN-backward reduction: a multiple of the channel-group count so that the grid stride
// keeps every thread on one channel group.
inline long bn_bwd_blocks_for(long pixels, int C, bool vec) {
  const int CG = vec ? C / 4 : C;
  const long items = pixels * CG;
  long want = (items + kThreads * 16L - 1) / (kThreads * 16L);
  if (want > 2048) want = 2048;
  if (want < 1) want = 1;
  long blocks = ((want + CG - 1) / CG) * CG;
  return blocks;
}
}  // namespace

namespace {
// Unused rows of a partial-sum workspace are zeroed by a KERNEL, not by hipMemsetAsync.  History: round 4 saw non-finite
// BatchNorm gradients from the second replay of a captured training step and blamed the ordering of memset NODES; round 5
// read the captured graph back (tools/probes/graph_topology.py): memset nodes sit in the chain like any kernel node and
// the symptom did not reproduce once warm-up and capture shared one stream.  The kernel stays -- it
// costs the same and keeps the captured step a chain of kernel nodes only; unetpp_debug_set("MEMSET_NODES", 1) switches
// to the memset for that probe.
__global__ __launch_bounds__(kThreads) void zero_rows_kernel(float* __restrict__ p, long n) {
  for (long i = blockIdx.x * static_cast<long>(kThreads) + threadIdx.x; i < n; i += static_cast<long>(gridDim.x) * kThreads)
    p[i] = 0.f;
}
inline void zero_rows(float* p, long n, hipStream_t st) {
  if (opt_value(OPT_MEMSET_NODES, 0) == 1) {   // tools/probes/graph_topology.py only: what a memset NODE does to a captured step
    (void)hipMemsetAsync(p, 0, static_cast<size_t>(n) * sizeof(float), st);
    return;
  }
  const long want = (n + kThreads - 1) / kThreads;
  hipLaunchKernelGGL(zero_rows_kernel, dim3(static_cast<unsigned>(want < 1024 ? want : 1024)), dim3(kThreads), 0, st, p, n);
}
}  // namespace

extern "C" int64_t unetpp_bn_bwd_blocks(int64_t pixels, int32_t C) {
  if (pixels < 1 || C < 1) return 0;
  // upper bound over both code paths (vector / scalar) so one workspace size serves either
  const long a = (C % 4 == 0) ? bn_bwd_blocks_for(pixels, C, true) : 0;
  const long b = bn_bwd_blocks_for(pixels, C, false);
  return a > b ? a : b;
}

extern "C" int unetpp_bn_bwd_reduce(const float* d_act, const float* y, const float* scale, const float* shift,
                                    const float* mean, const float* invstd, int64_t pixels, int32_t C, float* partial,
                                    void* stream) {
  if (!d_act || !y || !scale || !shift || !mean || !invstd || !partial || pixels < 1 || C < 1) return UNETPP_EINVAL;
  const bool vec = (C % 4 == 0) && aligned16(d_act) && aligned16(y);
  // the partial buffer always has unetpp_bn_bwd_blocks() rows; rows beyond this launch's grid are zeroed
  const long rows = unetpp_bn_bwd_blocks(pixels, C);
  const long blocks = bn_bwd_blocks_for(pixels, C, vec);
  if (blocks < rows) zero_rows(partial + blocks * C * 2, (rows - blocks) * C * 2, ST(stream));
  if (vec)
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<4>, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, ST(stream),
                       d_act, y, scale, shift, mean, invstd, pixels * (C / 4), C / 4, partial);
  else
    hipLaunchKernelGGL(bn_bwd_reduce_kernel<1>, dim3(static_cast<unsigned>(blocks)), dim3(kThreads), 0, ST(stream),
                       d_act, y, scale, shift, mean, invstd, pixels * C, C, partial);
  return launch_status();
}

extern "C" int unetpp_bn_bwd_finalize(const float* partial, int64_t n_blocks, int32_t C, float* dgamma, float* dbeta,
                                      void* stream) {
  if (!partial || n_blocks < 1 || C < 1 || !dgamma || !dbeta) return UNETPP_EINVAL;
  if ((reinterpret_cast<uintptr_t>(partial) & 7) != 0) return UNETPP_EINVAL;
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(fin_threads(n_blocks)), 0, ST(stream), partial, n_blocks, C,
                     dgamma, dbeta);
  return launch_status();
}

extern "C" int unetpp_bn_bwd_apply(const float* d_act, const float* y, const float* scale, const float* shift,
                                   const float* mean, const float* invstd, const float* gamma, const float* dgamma,
                                   const float* dbeta, int64_t pixels, int32_t C, float* dy, void* stream) {
  if (!d_act || !y || !scale || !shift || !mean || !invstd || !gamma || !dgamma || !dbeta || !dy || pixels < 1 || C < 1)
    return UNETPP_EINVAL;
  const bool vec = (C % 4 == 0) && aligned16(d_act) && aligned16(y) && aligned16(dy);
  const float inv_count = 1.0f / static_cast<float>(pixels);
  if (vec) {
    const long items = pixels * (C / 4);
    hipLaunchKernelGGL(bn_bwd_apply_kernel<4>, dim3(grid_for(items)), dim3(kThreads), 0, ST(stream), d_act, y, scale,
                       shift, mean, invstd, gamma, dgamma, dbeta, inv_count, items, C / 4, dy);
  } else {
    const long items = pixels * C;
    hipLaunchKernelGGL(bn_bwd_apply_kernel<1>, dim3(grid_for(items)), dim3(kThreads), 0, ST(stream), d_act, y, scale,
                       shift, mean, invstd, gamma, dgamma, dbeta, inv_count, items, C, dy);
  }
  return launch_status();
}

namespace {
// eligibility of the pool-routing BatchNorm backward (see bn_bwd_reduce_pool_kernel)
inline bool bn_bwd_pool_ok(int N, int H, int W, int C) {
  if (N < 1 || H < 2 || W < 2 || C < 4 || (H & 1) || (W & 1) || (C & 3)) return false;
  const int CG = C >> 2;
  if ((CG & (CG - 1)) != 0 || CG > kThreads) return false;
  return static_cast<long>(N) * H * W * C < 0x7fffffffL;
}
inline unsigned ilog2(unsigned v) {
  unsigned l = 0;
  while ((1u << l) < v) ++l;
  return l;
}
}  // namespace

extern "C" int unetpp_bn_bwd_pool_ok(int32_t N, int32_t H, int32_t W, int32_t C) { return bn_bwd_pool_ok(N, H, W, C) ? 1 : 0; }

extern "C" int unetpp_bn_bwd_reduce_pool(const float* d_act, const float* y, const float* scale, const float* shift,
                                         const float* mean, const float* invstd, const float* d_pooled,
                                         const uint8_t* pool_idx, int32_t N, int32_t H, int32_t W, int32_t C,
                                         float* partial, void* stream) {
  if (!d_act || !y || !scale || !shift || !mean || !invstd || !d_pooled || !pool_idx || !partial) return UNETPP_EINVAL;
  if (!bn_bwd_pool_ok(N, H, W, C)) return UNETPP_EINVAL;
  if (!aligned16(d_act) || !aligned16(y) || !aligned16(scale) || !aligned16(shift) || !aligned16(mean) ||
      !aligned16(invstd) || !aligned16(d_pooled) || (reinterpret_cast<uintptr_t>(pool_idx) & 3) != 0)
    return UNETPP_EINVAL;
  const long pixels = static_cast<long>(N) * H * W;
  const long rows_buf = unetpp_bn_bwd_blocks(pixels, C);  // rows of the partial buffer; unused ones are zeroed
  const long img_rows = static_cast<long>(N) * H;
  const long grid = img_rows < rows_buf ? img_rows : rows_buf;
  if (grid < rows_buf) zero_rows(partial + grid * C * 2, (rows_buf - grid) * C * 2, ST(stream));
  hipLaunchKernelGGL(bn_bwd_reduce_pool_kernel, dim3(static_cast<unsigned>(grid)), dim3(kThreads), 0, ST(stream),
                     reinterpret_cast<const f32x4*>(d_act), reinterpret_cast<const f32x4*>(y),
                     reinterpret_cast<const f32x4*>(scale), reinterpret_cast<const f32x4*>(shift),
                     reinterpret_cast<const f32x4*>(mean), reinterpret_cast<const f32x4*>(invstd),
                     reinterpret_cast<const f32x4*>(d_pooled), reinterpret_cast<const uint32_t*>(pool_idx),
                     static_cast<unsigned>(img_rows), static_cast<unsigned>(W), ilog2(static_cast<unsigned>(C >> 2)),
                     partial);
  return launch_status();
}

extern "C" int unetpp_bn_bwd_apply_pool(const float* d_act, const float* y, const float* scale, const float* shift,
                                        const float* mean, const float* invstd, const float* gamma, const float* dgamma,
                                        const float* dbeta, const float* d_pooled, const uint8_t* pool_idx, int32_t N,
                                        int32_t H, int32_t W, int32_t C, float* dy, void* stream) {
  if (!d_act || !y || !scale || !shift || !mean || !invstd || !gamma || !dgamma || !dbeta || !d_pooled || !pool_idx || !dy)
    return UNETPP_EINVAL;
  if (!bn_bwd_pool_ok(N, H, W, C)) return UNETPP_EINVAL;
  if (!aligned16(d_act) || !aligned16(y) || !aligned16(dy) || !aligned16(scale) || !aligned16(shift) ||
      !aligned16(mean) || !aligned16(invstd) || !aligned16(gamma) || !aligned16(dgamma) || !aligned16(dbeta) ||
      !aligned16(d_pooled) || (reinterpret_cast<uintptr_t>(pool_idx) & 3) != 0)
    return UNETPP_EINVAL;
  const long img_rows = static_cast<long>(N) * H;
  const float inv_count = 1.0f / static_cast<float>(img_rows * W);
  hipLaunchKernelGGL(bn_bwd_apply_pool_kernel, dim3(static_cast<unsigned>(img_rows < 16384 ? img_rows : 16384)),
                     dim3(kThreads), 0, ST(stream), reinterpret_cast<const f32x4*>(d_act),
                     reinterpret_cast<const f32x4*>(y), reinterpret_cast<const f32x4*>(scale),
                     reinterpret_cast<const f32x4*>(shift), reinterpret_cast<const f32x4*>(mean),
                     reinterpret_cast<const f32x4*>(invstd), reinterpret_cast<const f32x4*>(gamma),
                     reinterpret_cast<const f32x4*>(dgamma), reinterpret_cast<const f32x4*>(dbeta),
                     reinterpret_cast<const f32x4*>(d_pooled), reinterpret_cast<const uint32_t*>(pool_idx), inv_count,
                     static_cast<unsigned>(img_rows), static_cast<unsigned>(W), ilog2(static_cast<unsigned>(C >> 2)),
                     reinterpret_cast<f32x4*>(dy));
  return launch_status();
}

namespace {
inline bool head_args_ok(int N, int H, int W, int C, int n_cls, float p_drop) {
  return N >= 1 && H >= 1 && W >= 1 && C >= 1 && C <= kHeadMaxC && n_cls >= 1 && n_cls <= kHeadMaxCls &&
         p_drop >= 0.f && p_drop < 1.f;
}
}  // namespace

extern "C" int unetpp_head_fwd(const float* x, const float* weight, const float* bias, int32_t N, int32_t H, int32_t W,
                               int32_t C, int32_t n_cls, float p_drop, uint64_t seed, const uint8_t* mask, const uint64_t* seed_dev,
                               float* out_nchw, void* stream) {
  if (!x || !weight || !bias || !out_nchw || !head_args_ok(N, H, W, C, n_cls, p_drop)) return UNETPP_EINVAL;
  const long pixels = static_cast<long>(N) * H * W;
  const int use_drop = p_drop > 0.f;
  const int g4 = C >> 2;
  const int pcls = n_cls <= 4 ? 4 : 8;
  if ((C & 3) == 0 && aligned16(x) && aligned16(weight) && (g4 & (g4 - 1)) == 0 && g4 <= 32 && pcls <= g4 && pixels * C < 0x7fffffffL &&
      (mask == nullptr || (reinterpret_cast<uintptr_t>(mask) & 3) == 0)) {
    const long ppb = kThreads / g4;
    const long want = (pixels + ppb - 1) / ppb;
    const dim3 grid(static_cast<unsigned>(want < 256 * 16 ? want : 256 * 16));
#define UNETPP_HEAD_STREAM_D(L, PC, D)                                                                              \
  hipLaunchKernelGGL((head_fwd_stream_kernel<L, PC, D>), grid, dim3(kThreads), 0, ST(stream), x, weight, bias,         \
                     static_cast<unsigned>(pixels), static_cast<unsigned>(H * W), n_cls, 1.0f / (1.0f - p_drop),        \
                     keep_threshold(p_drop), seed, mask, seed_dev, out_nchw)
#define UNETPP_HEAD_STREAM(L, PC)                              \
  do {                                                         \
    if (!use_drop) UNETPP_HEAD_STREAM_D(L, PC, 0);             \
    else if (mask == nullptr) UNETPP_HEAD_STREAM_D(L, PC, 1);  \
    else UNETPP_HEAD_STREAM_D(L, PC, 2);                       \
  } while (0)
    if (pcls == 4) {
      switch (g4) {
        case 4: UNETPP_HEAD_STREAM(2, 4); break;
        case 8: UNETPP_HEAD_STREAM(3, 4); break;
        case 16: UNETPP_HEAD_STREAM(4, 4); break;
        default: UNETPP_HEAD_STREAM(5, 4); break;
      }
    } else {
      switch (g4) {
        case 8: UNETPP_HEAD_STREAM(3, 8); break;
        case 16: UNETPP_HEAD_STREAM(4, 8); break;
        default: UNETPP_HEAD_STREAM(5, 8); break;
      }
    }
#undef UNETPP_HEAD_STREAM
#undef UNETPP_HEAD_STREAM_D
    return launch_status();
  }
  if ((C & 3) == 0 && aligned16(x)) {
    const long tiles = (pixels + 63) / 64;
    const unsigned blocks = static_cast<unsigned>(tiles < 256 * 16 ? tiles : 256 * 16);
    hipLaunchKernelGGL(head_fwd_tiled_kernel, dim3(blocks), dim3(64), 64 * (C + 1) * sizeof(float), ST(stream), x, weight,
                       bias, pixels, H * W, C,
                       n_cls, 1.0f / (1.0f - p_drop), keep_threshold(p_drop), seed, mask, seed_dev, use_drop, out_nchw);
    return launch_status();
  }
  hipLaunchKernelGGL(head_fwd_kernel, dim3(grid_for(pixels)), dim3(kThreads), 0, ST(stream), x, weight, bias, pixels,
                     H * W, C, n_cls, 1.0f / (1.0f - p_drop), keep_threshold(p_drop), seed, mask, seed_dev, use_drop, out_nchw);
  return launch_status();
}

extern "C" int64_t unetpp_head_bwd_blocks(int64_t pixels) {
  if (pixels < 1) return 0;
  const long tiles = (pixels + 63) / 64;
  return tiles < 4096 ? tiles : 4096;  // 16 workgroups per CU: the tile loop is a chain of dependent loads, occupancy hides it
}

extern "C" int unetpp_head_bwd(const float* d_out_nchw, const float* out_nchw, const float* x, const float* weight,
                               int32_t N, int32_t H, int32_t W, int32_t C, int32_t n_cls, float p_drop, uint64_t seed,
                               const uint8_t* mask, const uint64_t* seed_dev, float* dx, int32_t accumulate, int32_t gate_x, float* partial,
                               void* stream) {
  if (!d_out_nchw || !out_nchw || !x || !weight || !dx || !partial || !head_args_ok(N, H, W, C, n_cls, p_drop))
    return UNETPP_EINVAL;
  const long pixels = static_cast<long>(N) * H * W;
  const int use_drop = p_drop > 0.f;
  const int g4 = C >> 2;
  if ((C & 3) == 0 && aligned16(x) && aligned16(dx) && aligned16(weight) && (g4 & (g4 - 1)) == 0 && g4 >= 2 && g4 <= 32 &&
      pixels * C < 0x7fffffffL && (mask == nullptr || (reinterpret_cast<uintptr_t>(mask) & 3) == 0)) {
    const size_t lds = (64 * (C + 1) + 64 * kHeadMaxCls + 2 * n_cls * C) * sizeof(float);
    const dim3 grid(static_cast<unsigned>(unetpp_head_bwd_blocks(pixels)));
    const int drop = !use_drop ? 0 : (mask == nullptr ? 1 : 2);
#define UNETPP_HEAD_BWD(L, D)                    \
  do {                                           \
    if (n_cls <= 4) UNETPP_HEAD_BWD_P(L, D, 4);  \
    else UNETPP_HEAD_BWD_P(L, D, 8);             \
  } while (0)
#define UNETPP_HEAD_BWD_P(L, D, PC)                                                                                 \
  hipLaunchKernelGGL((head_bwd_pow2_kernel<L, D, PC>), grid, dim3(kThreads), lds, ST(stream), d_out_nchw, out_nchw, x, \
                     weight, static_cast<unsigned>(pixels), static_cast<unsigned>(H * W), n_cls, 1.0f / (1.0f - p_drop), \
                     keep_threshold(p_drop), seed, mask, seed_dev, dx, accumulate, gate_x, partial)
#define UNETPP_HEAD_BWD_L(L)              \
  do {                                    \
    if (drop == 0) UNETPP_HEAD_BWD(L, 0); \
    else if (drop == 1) UNETPP_HEAD_BWD(L, 1); \
    else UNETPP_HEAD_BWD(L, 2);           \
  } while (0)
    switch (g4) {
      case 2: UNETPP_HEAD_BWD_L(1); break;
      case 4: UNETPP_HEAD_BWD_L(2); break;
      case 8: UNETPP_HEAD_BWD_L(3); break;
      case 16: UNETPP_HEAD_BWD_L(4); break;
      default: UNETPP_HEAD_BWD_L(5); break;
    }
#undef UNETPP_HEAD_BWD_L
#undef UNETPP_HEAD_BWD
#undef UNETPP_HEAD_BWD_P
    return launch_status();
  }
  if ((C & 3) == 0 && aligned16(x) && aligned16(dx)) {
    const size_t lds = (64 * (C + 1) + 64 * kHeadMaxCls + kHeadMaxCls * C + 2 * n_cls * C) * sizeof(float);
    hipLaunchKernelGGL(head_bwd_vec_kernel, dim3(static_cast<unsigned>(unetpp_head_bwd_blocks(pixels))), dim3(kThreads),
                       lds, ST(stream), d_out_nchw, out_nchw, x, weight, pixels, H * W, C, n_cls, 1.0f / (1.0f - p_drop),
                       keep_threshold(p_drop), seed, mask, seed_dev, use_drop, dx, accumulate, gate_x, partial);
    return launch_status();
  }
  hipLaunchKernelGGL(head_bwd_kernel, dim3(static_cast<unsigned>(unetpp_head_bwd_blocks(pixels))), dim3(kThreads), 0,
                     ST(stream), d_out_nchw, out_nchw, x, weight, pixels, H * W, C, n_cls, 1.0f / (1.0f - p_drop),
                     keep_threshold(p_drop), seed, mask, seed_dev, use_drop, dx, accumulate, gate_x, partial);
  return launch_status();
}

extern "C" int unetpp_sum_partials(const float* partial, int64_t n_blocks, int64_t len, float* out, void* stream) {
  if (!partial || !out || n_blocks < 1 || len < 1) return UNETPP_EINVAL;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(static_cast<unsigned>((len + 15) / 16)), dim3(1024), 0, ST(stream),
                     partial, n_blocks, len, out);
  return launch_status();
}

extern "C" int unetpp_bilinear2x_fwd(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, float* y, void* stream) {
  if (!x || !y || N < 1 || H < 1 || W < 1 || C < 1) return UNETPP_EINVAL;
  const long items = static_cast<long>(N) * 4 * H * W * C;
  hipLaunchKernelGGL(bilinear2x_fwd_kernel, dim3(grid_for(items)), dim3(kThreads), 0, ST(stream), x, N, H, W, C, y);
  return launch_status();
}

extern "C" int unetpp_bilinear2x_bwd(const float* dy, int32_t N, int32_t H, int32_t W, int32_t C, float* dx,
                                     int32_t accumulate, void* stream) {
  if (!dy || !dx || N < 1 || H < 1 || W < 1 || C < 1) return UNETPP_EINVAL;
  const long items = static_cast<long>(N) * H * W * C;
  hipLaunchKernelGGL(bilinear2x_bwd_kernel, dim3(grid_for(items)), dim3(kThreads), 0, ST(stream), dy, N, H, W, C, dx, accumulate);
  return launch_status();
}

extern "C" int unetpp_nchw_to_nhwc(const float* src, int32_t N, int32_t C, int32_t H, int32_t W, float* dst, void* stream) {
  if (!src || !dst || N < 1 || C < 1 || H < 1 || W < 1) return UNETPP_EINVAL;
  const long items = static_cast<long>(N) * C * H * W;
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(grid_for(items)), dim3(kThreads), 0, ST(stream), src, N, C,
                     static_cast<long>(H) * W, dst);
  return launch_status();
}

extern "C" int unetpp_nhwc_to_nchw(const float* src, int32_t N, int32_t C, int32_t H, int32_t W, float* dst, void* stream) {
  if (!src || !dst || N < 1 || C < 1 || H < 1 || W < 1) return UNETPP_EINVAL;
  const long items = static_cast<long>(N) * C * H * W;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(grid_for(items)), dim3(kThreads), 0, ST(stream), src, N, C,
                     static_cast<long>(H) * W, dst);
  return launch_status();
}

extern "C" int unetpp_abi_version(void) { return UNETPP_ABI_VERSION; }
extern "C" const char* unetpp_build_arch(void) { return "gfx950"; }

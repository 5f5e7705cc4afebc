// Caller-side kernels of the training step: the loss and the target synthesis the reference runs on
// the CPU each step (trainer/trainer.py:122-135; both carry a "put on GPU" TODO there).
//   * FocalLoss_BCE_2d (tools/losses/focal_loss.py:255-301), forward and gradient in one pass over (pred, target):
//       e = 1 - |p - t| + 1e-20,  L = sum -(1 - e)^gamma log e / rows,  dL/dp = [u^gamma / e - gamma u^(gamma-1) log e] sign(p - t) / rows
//     (u = 1 - e; abs'(0) = 0 as in autograd).  Per-block fp32 partial sums, fixed-order finish (unetpp_sum_partials).
//   * create_heatmap (tools/misc/helper.py:87-172): Gaussian-of-distance maps exp(-0.5 sqrt(dx^2 + dy^2) / R) of the
//     key points, channels 1 and 3 divided by their per-image maximum.  Each map is evaluated in fp64; the sums of channels 1 and 3
//     are float32 running sums and the quotient is float32, as in the reference's numpy code.
#include "common.h"

namespace unetpp {
namespace {

constexpr int kLossThreads = 256;
constexpr int kLossPerThread = 8;  // elements per thread: 2 x float4

// The elements of one thread (2 x float4 at `base`): the gradient (times `scale`: 1 for a single head, 1 / heads under
// the trainer's mean over heads -- a second float32 product, as autograd forms it) and the thread's part of the loss sum.
// t[2][4]: the thread's targets, read once for all heads.
__device__ __forceinline__ float focal_thread(const float* __restrict__ pred, float* __restrict__ grad,
                                              const float (&t)[kLossPerThread / 4][4], long base, long n, float gamma,
                                              float inv_rows, float scale) {
  const bool cube = gamma == 3.f;
  float sum = 0.f;
#pragma unroll
  for (int h = 0; h < kLossPerThread / 4; ++h) {
    const long i = base + 4 * h;
    if (i >= n) break;
    float p[4], g[4];
    const bool vec = (i + 4 <= n);
    if (vec) {
      const f32x4 pv = *reinterpret_cast<const f32x4*>(pred + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) p[e] = pv[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) p[e] = (i + e < n) ? pred[i + e] : 0.f;
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float d = p[e] - t[h][e];
      const float err = (1.f - fabsf(d)) + 1e-20f;
      const float u = 1.f - err;
      const float lg = logf(err);
      const float ug1 = cube ? u * u : powf(u, gamma - 1.f);  // u^(gamma-1)
      const float ug = ug1 * u;
      const float le = -ug * lg;
      const float dl_de = gamma * ug1 * lg - ug / err;
      const float sgn = (d > 0.f) ? 1.f : ((d < 0.f) ? -1.f : 0.f);
      g[e] = (-dl_de * sgn * inv_rows) * scale;
      if (i + e < n) sum += le;
    }
    if (grad != nullptr) {
      if (vec) {
        *reinterpret_cast<f32x4*>(grad + i) = f32x4{g[0], g[1], g[2], g[3]};
      } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
          if (i + e < n) grad[i + e] = g[e];
      }
    }
  }
  return sum;
}

__device__ __forceinline__ void focal_read_target(const float* __restrict__ target, long base, long n,
                                                  float (&t)[kLossPerThread / 4][4]) {
#pragma unroll
  for (int h = 0; h < kLossPerThread / 4; ++h) {
    const long i = base + 4 * h;
    if (i + 4 <= n) {
      const f32x4 tv = *reinterpret_cast<const f32x4*>(target + i);
#pragma unroll
      for (int e = 0; e < 4; ++e) t[h][e] = tv[e];
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e) t[h][e] = (i + e < n) ? target[i + e] : 0.f;
    }
  }
}

__global__ __launch_bounds__(kLossThreads) void focal_bce_kernel(const float* __restrict__ pred,
                                                                const float* __restrict__ target, long n, float gamma,
                                                                float inv_rows, float* __restrict__ grad,
                                                                float* __restrict__ partial) {
  __shared__ float red[kLossThreads / 64];
  const long base = (blockIdx.x * static_cast<long>(kLossThreads) + threadIdx.x) * kLossPerThread;
  float t[kLossPerThread / 4][4];
  focal_read_target(target, base, n, t);
  float sum = focal_thread(pred, grad, t, base, n, gamma, inv_rows, 1.f);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = sum;
  __syncthreads();
  if (threadIdx.x == 0) partial[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) * inv_rows;
}

// All deep-supervision heads against one target: the target is read once, partial[h * gridDim.x + block] as
// focal_bce_kernel writes partial[block] for head h (same per-thread order, same wave and block sums).
__global__ __launch_bounds__(kLossThreads) void focal_bce_heads_kernel(const unetpp_focal_heads hd,
                                                                      const float* __restrict__ target, long n, float gamma,
                                                                      float inv_rows, float scale,
                                                                      float* __restrict__ partial) {
  __shared__ float red[UNETPP_MAX_HEADS][kLossThreads / 64];
  const long base = (blockIdx.x * static_cast<long>(kLossThreads) + threadIdx.x) * kLossPerThread;
  float t[kLossPerThread / 4][4];
  focal_read_target(target, base, n, t);
  for (int h = 0; h < hd.n_heads; ++h) {
    float sum = focal_thread(hd.pred[h], hd.grad[h], t, base, n, gamma, inv_rows, scale);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    if ((threadIdx.x & 63) == 0) red[h][threadIdx.x >> 6] = sum;
  }
  __syncthreads();
  if (static_cast<int>(threadIdx.x) < hd.n_heads) {
    const float* r = red[threadIdx.x];
    partial[static_cast<long>(threadIdx.x) * gridDim.x + blockIdx.x] = ((r[0] + r[1]) + (r[2] + r[3])) * inv_rows;
  }
}

// one block: loss = sum of the per-block partials, fixed order (1024 strided sums, then an LDS tree)
__global__ __launch_bounds__(1024) void focal_bce_finish_kernel(const float* __restrict__ partial, long n_blocks,
                                                                float* __restrict__ loss) {
  __shared__ float red[1024];
  float s = 0.f;
  for (long i = threadIdx.x; i < n_blocks; i += 1024) s += partial[i];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int w = 512; w >= 1; w >>= 1) {
    if (static_cast<int>(threadIdx.x) < w) red[threadIdx.x] += red[threadIdx.x + w];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = red[0];
}

// one block: every head's loss as focal_bce_finish_kernel sums it, then the trainer's mean over heads in ITS order:
// avg = 0; avg = avg + loss_h (h = 0, 1, ...); avg = 1.0 * avg / heads -- float32 tensor arithmetic, the division by a
// Python scalar being a product with float32(1 / heads).
__global__ __launch_bounds__(1024) void focal_bce_heads_finish_kernel(const float* __restrict__ partial, long n_blocks,
                                                                      int n_heads, float inv_heads, float* __restrict__ loss) {
  __shared__ float red[1024];
  float avg = 0.f;
  for (int h = 0; h < n_heads; ++h) {
    float s = 0.f;
    for (long i = threadIdx.x; i < n_blocks; i += 1024) s += partial[h * n_blocks + i];
    __syncthreads();  // (the previous head's red[0] has been read)
    red[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w >= 1; w >>= 1) {
      if (static_cast<int>(threadIdx.x) < w) red[threadIdx.x] += red[threadIdx.x + w];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      loss[1 + h] = red[0];
      avg = avg + red[0];
    }
  }
  if (threadIdx.x == 0) loss[0] = (1.0f * avg) * inv_heads;
}

// grid (blocks per image, N): unnormalised maps of the 4 channels + per-block maxima of channels 1 and 3
__global__ __launch_bounds__(256) void heatmap_kernel(const float* __restrict__ points, int P, int H, int W, double radius,
                                                      float* __restrict__ out, float* __restrict__ scratch,
                                                      float* __restrict__ blockmax) {
  __shared__ float red[2][4];
  const int n = blockIdx.y;
  const long hw = static_cast<long>(H) * W;
  const long i = blockIdx.x * 256L + threadIdx.x;
  const float* pts = points + static_cast<long>(n) * P * 2;
  double c0 = 0.0, c2 = 0.0;
  float c1 = 0.f, c3 = 0.f;  // channels 1 and 3: float32 running sums, rounded after every addition (the reference
                             // does `float32_array += float64_map`)
  if (i < hw) {
    const double y = static_cast<double>(i / W), x = static_cast<double>(i % W);
    for (int p = 0; p < P; ++p) {
      const double dx = x - static_cast<double>(pts[2 * p]), dy = y - static_cast<double>(pts[2 * p + 1]);
      const double v = exp(-0.5 * sqrt(dx * dx + dy * dy) / radius);
      if (p == 0) c0 = v;
      else if (p < 4) c1 = static_cast<float>(static_cast<double>(c1) + v);
      else if (p == 4) c2 = v;
      else c3 = static_cast<float>(static_cast<double>(c3) + v);
    }
    float* o = out + static_cast<long>(n) * 4 * hw + i;
    o[0] = static_cast<float>(c0);
    o[2 * hw] = static_cast<float>(c2);
    float* s = scratch + static_cast<long>(n) * 2 * hw + i;
    s[0] = c1;
    s[hw] = c3;
  }
  float m1 = c1, m3 = c3;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) {
    m1 = fmaxf(m1, __shfl_xor(m1, off));
    m3 = fmaxf(m3, __shfl_xor(m3, off));
  }
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = m1;
    red[1][threadIdx.x >> 6] = m3;
  }
  __syncthreads();
  if (threadIdx.x < 2)
    blockmax[(static_cast<long>(n) * 2 + threadIdx.x) * gridDim.x + blockIdx.x] =
        fmaxf(fmaxf(red[threadIdx.x][0], red[threadIdx.x][1]), fmaxf(red[threadIdx.x][2], red[threadIdx.x][3]));
}

__global__ __launch_bounds__(256) void heatmap_norm_kernel(int H, int W, float* __restrict__ out,
                                                           const float* __restrict__ scratch,
                                                           const float* __restrict__ blockmax) {
  __shared__ float red[2][256];
  const int n = blockIdx.y;
  const long hw = static_cast<long>(H) * W;
  float m1 = 0.f, m3 = 0.f;
  for (unsigned b = threadIdx.x; b < gridDim.x; b += 256) {
    m1 = fmaxf(m1, blockmax[(static_cast<long>(n) * 2 + 0) * gridDim.x + b]);
    m3 = fmaxf(m3, blockmax[(static_cast<long>(n) * 2 + 1) * gridDim.x + b]);
  }
  red[0][threadIdx.x] = m1;
  red[1][threadIdx.x] = m3;
  __syncthreads();
  for (int s = 128; s >= 1; s >>= 1) {
    if (static_cast<int>(threadIdx.x) < s) {
      red[0][threadIdx.x] = fmaxf(red[0][threadIdx.x], red[0][threadIdx.x + s]);
      red[1][threadIdx.x] = fmaxf(red[1][threadIdx.x], red[1][threadIdx.x + s]);
    }
    __syncthreads();
  }
  const long i = blockIdx.x * 256L + threadIdx.x;
  if (i >= hw) return;
  const float* s = scratch + static_cast<long>(n) * 2 * hw + i;
  float* o = out + static_cast<long>(n) * 4 * hw + i;
  o[hw] = s[0] / red[0][0];  // float32 sum / float32 maximum, as the reference's array arithmetic
  o[3 * hw] = s[hw] / red[1][0];
}

}  // namespace
}  // namespace unetpp

using namespace unetpp;

extern "C" int64_t unetpp_focal_bce_blocks(int64_t n) {
  if (n < 1) return 0;
  const int64_t per_block = static_cast<int64_t>(kLossThreads) * kLossPerThread;
  return (n + per_block - 1) / per_block;
}

extern "C" int unetpp_focal_bce(const float* pred, const float* target, int64_t n, int64_t rows, float gamma, float* grad,
                                float* partial, float* loss, void* stream) {
  if (pred == nullptr || target == nullptr || partial == nullptr || loss == nullptr || n < 1 || rows < 1) return UNETPP_EINVAL;
  if (((reinterpret_cast<uintptr_t>(pred) | reinterpret_cast<uintptr_t>(target) | reinterpret_cast<uintptr_t>(grad)) & 15) != 0)
    return UNETPP_EINVAL;
  const int64_t blocks = unetpp_focal_bce_blocks(n);
  if (blocks > 0x7fffffffLL) return UNETPP_EINVAL;
  hipLaunchKernelGGL(focal_bce_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kLossThreads), 0,
                     static_cast<hipStream_t>(stream), pred, target, static_cast<long>(n), gamma,
                     1.f / static_cast<float>(rows), grad, partial);
  hipLaunchKernelGGL(focal_bce_finish_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), partial,
                     static_cast<long>(blocks), loss);
  return launch_status();
}

extern "C" int unetpp_focal_bce_heads(const unetpp_focal_heads* heads, const float* target, int64_t n, int64_t rows,
                                      float gamma, float* partial, float* loss, void* stream) {
  if (heads == nullptr || target == nullptr || partial == nullptr || loss == nullptr || n < 1 || rows < 1) return UNETPP_EINVAL;
  if (heads->n_heads < 1 || heads->n_heads > UNETPP_MAX_HEADS) return UNETPP_EINVAL;
  uintptr_t bits = reinterpret_cast<uintptr_t>(target);
  for (int h = 0; h < heads->n_heads; ++h) {
    if (heads->pred[h] == nullptr) return UNETPP_EINVAL;
    bits |= reinterpret_cast<uintptr_t>(heads->pred[h]) | reinterpret_cast<uintptr_t>(heads->grad[h]);
  }
  if ((bits & 15) != 0) return UNETPP_EINVAL;
  const int64_t blocks = unetpp_focal_bce_blocks(n);
  if (blocks > 0x7fffffffLL) return UNETPP_EINVAL;
  unetpp_focal_heads hd = *heads;
  for (int h = hd.n_heads; h < UNETPP_MAX_HEADS; ++h) hd.pred[h] = nullptr, hd.grad[h] = nullptr;
  const float inv_heads = 1.0f / static_cast<float>(hd.n_heads);
  hipLaunchKernelGGL(focal_bce_heads_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kLossThreads), 0,
                     static_cast<hipStream_t>(stream), hd, target, static_cast<long>(n), gamma,
                     1.f / static_cast<float>(rows), inv_heads, partial);
  hipLaunchKernelGGL(focal_bce_heads_finish_kernel, dim3(1), dim3(1024), 0, static_cast<hipStream_t>(stream), partial,
                     static_cast<long>(blocks), hd.n_heads, inv_heads, loss);
  return launch_status();
}

extern "C" int64_t unetpp_heatmap_workspace_bytes(int32_t N, int32_t H, int32_t W) {
  if (N < 1 || H < 1 || W < 1) return 0;
  const int64_t hw = static_cast<int64_t>(H) * W, blocks = (hw + 255) / 256;
  return (static_cast<int64_t>(N) * 2 * hw + static_cast<int64_t>(N) * 2 * blocks) * 4;
}

extern "C" int unetpp_create_heatmap(const float* points, int32_t N, int32_t P, int32_t H, int32_t W, float radius,
                                     float* out_nchw, void* workspace, void* stream) {
  if (points == nullptr || out_nchw == nullptr || workspace == nullptr) return UNETPP_EINVAL;
  if (N < 1 || N > 65535 || P < 6 || H < 1 || W < 1 || !(radius > 0.f)) return UNETPP_EINVAL;
  const long hw = static_cast<long>(H) * W;
  const unsigned blocks = static_cast<unsigned>((hw + 255) / 256);
  float* scratch = static_cast<float*>(workspace);
  float* blockmax = scratch + static_cast<long>(N) * 2 * hw;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipLaunchKernelGGL(heatmap_kernel, dim3(blocks, N), dim3(256), 0, st, points, P, H, W, static_cast<double>(radius),
                     out_nchw, scratch, blockmax);
  hipLaunchKernelGGL(heatmap_norm_kernel, dim3(blocks, N), dim3(256), 0, st, H, W, out_nchw, scratch, blockmax);
  return launch_status();
}

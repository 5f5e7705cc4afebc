// Heat-map side of validation on the device (/root/reference/tools/misc/heatmap.py):
//   unetpp_heatmap_pattern     pattern-driven target maps (heatmap.py:203-230): per map the Gaussians
//                              exp(-0.5 * distance / radius) of its key points summed in float32 (one rounding per
//                              addition, in pattern order), divided by the map's maximum
//   unetpp_keypoints_extract   heat map -> up to `num` (x, y) peaks per map (heatmap.py:148-200): values below the
//                              threshold zeroed, 3x3 median > 0 as the region mask, the mask cut into regions, regions
//                              ordered by their maximum (descending, ties in raster order of the region's first pixel),
//                              each reported at the first pixel in raster order that attains it.
// Regions, as the reference finds them with OpenCV (region_segment_, heatmap.py:100-144), restated from the published
// algorithms of the calls it makes (stages 3-8 below): the 3x3 chamfer distance transform in 16-bit fixed point, cores =
// distance > float32(0.1 * the map's maximum), 8-connected components of the cores, and the watershed of the BINARY mask
// image seeded with them: cores grow over 4-neighbours through their blob, the background marker through the two-pixel
// ring the reference leaves unknown around the mask, a pixel reached by two different labels in the same round becomes a
// line.  Blobs that touch are split, a blob without a core (thinner than a tenth of the thickest) is no region.  The
// round-2 stand-in -- a region = an 8-connected component of the mask -- stays available (stages 0, 1, 2 alone).
// OpenCV is absent from the build image, so this row is "parity unpinned": checked against oracle/keypoints_oracle.py only.
//
// All kernels are HBM/latency trivial (a 512x512 map is 1 MB); what matters is that the whole extraction stays on the
// device: the reference moves every head output to the CPU and runs OpenCV per map (trainer/trainer.py:213-221).
// Labels: label[p] = smallest raster index of p's component, by min-propagation over the 8 neighbours followed by
// pointer jumping, iterated until a sweep changes nothing (device flag read by the host every few sweeps).  The only
// atomics are atomicMin/atomicMax on integers (order independent: results are bitwise reproducible).
#include "common.h"

namespace unetpp {
namespace {

constexpr int kKpThreads = 256;

// ---- pattern maps: thread = pixel, blockIdx.y = (image, map) ----
__global__ __launch_bounds__(kKpThreads) void pattern_map_kernel(const float* __restrict__ points, int P,
                                                                 const int* __restrict__ map_points,
                                                                 const int* __restrict__ map_begin, int n_maps, int H,
                                                                 int W, double radius, float* __restrict__ out,
                                                                 float* __restrict__ blockmax) {
  __shared__ float red[kKpThreads / 64];
  const int n = blockIdx.y / n_maps, m = blockIdx.y % n_maps;
  const long hw = static_cast<long>(H) * W;
  const long i = blockIdx.x * static_cast<long>(kKpThreads) + threadIdx.x;
  const float* pts = points + static_cast<long>(n) * P * 2;
  float acc = 0.f;
  if (i < hw) {
    const double y = static_cast<double>(i / W), x = static_cast<double>(i % W);
    for (int k = map_begin[m]; k < map_begin[m + 1]; ++k) {
      const int p = map_points[k];
      const double dx = x - static_cast<double>(pts[2 * p]), dy = y - static_cast<double>(pts[2 * p + 1]);
      acc = static_cast<float>(static_cast<double>(acc) + exp(-0.5 * sqrt(dx * dx + dy * dy) / radius));
    }
    out[static_cast<long>(blockIdx.y) * hw + i] = acc;
  }
  float mx = acc;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mx;
  __syncthreads();
  if (threadIdx.x == 0)
    blockmax[static_cast<long>(blockIdx.y) * gridDim.x + blockIdx.x] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

__global__ __launch_bounds__(kKpThreads) void pattern_norm_kernel(int H, int W, float* __restrict__ out,
                                                                  const float* __restrict__ blockmax) {
  __shared__ float red[kKpThreads];
  const long hw = static_cast<long>(H) * W;
  float mx = 0.f;
  for (unsigned b = threadIdx.x; b < gridDim.x; b += kKpThreads)
    mx = fmaxf(mx, blockmax[static_cast<long>(blockIdx.y) * gridDim.x + b]);
  red[threadIdx.x] = mx;
  __syncthreads();
  for (int s = kKpThreads / 2; s >= 1; s >>= 1) {
    if (static_cast<int>(threadIdx.x) < s) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + s]);
    __syncthreads();
  }
  const long i = blockIdx.x * static_cast<long>(kKpThreads) + threadIdx.x;
  if (i < hw) out[static_cast<long>(blockIdx.y) * hw + i] /= red[0];  // float32 / float32, as the reference's arrays
}

// ---- extraction.  Work buffers per map: label int32 [H*W], best uint64 [H*W]. ----
__device__ __forceinline__ float thresholded(const float* heat, int H, int W, int y, int x, float thr) {
  y = min(max(y, 0), H - 1);  // replicated border (cv2.medianBlur's BORDER_REPLICATE)
  x = min(max(x, 0), W - 1);
  const float v = heat[static_cast<long>(y) * W + x];
  return v < thr ? 0.f : v;
}

// mask = (3x3 median of the thresholded map) > 0  <=>  at least 5 of the 9 window values are > 0; label = own index
__global__ __launch_bounds__(kKpThreads) void kp_mask_kernel(const float* __restrict__ heat, int H, int W,
                                                             const float* __restrict__ thr_per_map,
                                                             int* __restrict__ label, unsigned long long* __restrict__ best) {
  const long hw = static_cast<long>(H) * W;
  const long i = blockIdx.x * static_cast<long>(kKpThreads) + threadIdx.x;
  if (i >= hw) return;
  const float* hm = heat + static_cast<long>(blockIdx.y) * hw;
  const float thr = thr_per_map[blockIdx.y];
  const int y = static_cast<int>(i / W), x = static_cast<int>(i % W);
  int positive = 0;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) positive += thresholded(hm, H, W, y + dy, x + dx, thr) > 0.f ? 1 : 0;
  label[static_cast<long>(blockIdx.y) * hw + i] = positive >= 5 ? static_cast<int>(i) : -1;
  best[static_cast<long>(blockIdx.y) * hw + i] = 0ull;
}

// one sweep: every masked pixel pulls the smallest label of its 8 neighbours into its component's root
__global__ __launch_bounds__(kKpThreads) void kp_merge_kernel(int H, int W, int* __restrict__ label, int* __restrict__ changed) {
  const long hw = static_cast<long>(H) * W;
  const long i = blockIdx.x * static_cast<long>(kKpThreads) + threadIdx.x;
  if (i >= hw) return;
  int* lab = label + static_cast<long>(blockIdx.y) * hw;
  const int mine = lab[i];
  if (mine < 0) return;
  const int y = static_cast<int>(i / W), x = static_cast<int>(i % W);
  int m = mine;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int yy = y + dy, xx = x + dx;
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      const int l = lab[static_cast<long>(yy) * W + xx];
      if (l >= 0 && l < m) m = l;
    }
  if (m < mine) {
    atomicMin(&lab[mine], m);  // hook the old root under the smaller label
    atomicMin(&lab[i], m);
    *changed = 1;
  }
}

// pointer jumping: every pixel points at its root
__global__ __launch_bounds__(kKpThreads) void kp_flatten_kernel(int H, int W, int* __restrict__ label) {
  const long hw = static_cast<long>(H) * W;
  const long i = blockIdx.x * static_cast<long>(kKpThreads) + threadIdx.x;
  if (i >= hw) return;
  int* lab = label + static_cast<long>(blockIdx.y) * hw;
  int l = lab[i];
  if (l < 0) return;
  while (lab[l] != l) l = lab[l];
  lab[i] = l;
}

// ---- the reference's region step (heatmap.py:100-144).  Extra work buffers per map: dist int32 [H*W], marker int32 [H*W].
constexpr int kDistHV = 62587, kDistDiag = 89738;  // cv2.distanceTransform(DIST_L2, 3): 0.955 and 1.3693 in 16-bit fixed point
constexpr int kDistInf = 0x3FFFFFFF;

__global__ __launch_bounds__(kKpThreads) void kp_dist_init_kernel(int H, int W, const int* __restrict__ label,
                                                                  int* __restrict__ dist, int* __restrict__ mapmax) {
  const long hw = static_cast<long>(H) * W;
  const long i = blockIdx.x * static_cast<long>(kKpThreads) + threadIdx.x;
  if (i == 0) mapmax[blockIdx.y] = 0;
  if (i >= hw) return;
  dist[static_cast<long>(blockIdx.y) * hw + i] = label[static_cast<long>(blockIdx.y) * hw + i] >= 0 ? kDistInf : 0;
}

// one relaxation sweep of the chamfer distance: a pixel writes only its own entry and entries only ever decrease, so
// sweeps in any interleaving end at the one fixed point -- the shortest 8-neighbour path (weights HV / DIAG) to a
// non-mask pixel, which is what cv2's two sequential passes compute on a full rectangle.  Outside the image counts as
// far away (cv2 pads its work buffer with the maximum distance).
__global__ __launch_bounds__(kKpThreads) void kp_dist_sweep_kernel(int H, int W, int* __restrict__ dist, int* __restrict__ changed) {
  const long hw = static_cast<long>(H) * W;
  const long i = blockIdx.x * static_cast<long>(kKpThreads) + threadIdx.x;
  if (i >= hw) return;
  int* d = dist + static_cast<long>(blockIdx.y) * hw;
  const int mine = d[i];
  if (mine == 0) return;
  const int y = static_cast<int>(i / W), x = static_cast<int>(i % W);
  int m = mine;
#pragma unroll
  for (int dy = -1; dy <= 1; ++dy)
#pragma unroll
    for (int dx = -1; dx <= 1; ++dx) {
      const int yy = y + dy, xx = x + dx;
      if ((dy == 0 && dx == 0) || yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      const int v = __atomic_load_n(&d[static_cast<long>(yy) * W + xx], __ATOMIC_RELAXED) + ((dy != 0 && dx != 0) ? kDistDiag : kDistHV);
      m = v < m ? v : m;
    }
  if (m < mine) {
    __atomic_store_n(&d[i], m, __ATOMIC_RELAXED);
    *changed = 1;
  }
}

__global__ __launch_bounds__(kKpThreads) void kp_dist_max_kernel(int H, int W, const int* __restrict__ dist, int* __restrict__ mapmax) {
  __shared__ int red[kKpThreads / 64];
  const long hw = static_cast<long>(H) * W;
  const long i = blockIdx.x * static_cast<long>(kKpThreads) + threadIdx.x;
  int m = i < hw ? dist[static_cast<long>(blockIdx.y) * hw + i] : 0;
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) m = max(m, __shfl_xor(m, off));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  if (threadIdx.x == 0) {
    m = max(max(red[0], red[1]), max(red[2], red[3]));
    if (m > 0) atomicMax(&mapmax[blockIdx.y], m);
  }
}

// cores (the reference's sure_fg): distance as float32 (fixed point x 2^-16, cv2's output) above float32(0.1 * maximum),
// the product taken in double as Python takes it (heatmap.py:121).  label = own index on a core, -1 elsewhere: stage 1's
// component sweeps then run on the cores.
__global__ __launch_bounds__(kKpThreads) void kp_core_kernel(int H, int W, const int* __restrict__ dist,
                                                             const int* __restrict__ mapmax, int* __restrict__ label) {
  const long hw = static_cast<long>(H) * W;
  const long i = blockIdx.x * static_cast<long>(kKpThreads) + threadIdx.x;
  if (i >= hw) return;
  const float scale = 1.0f / 65536.0f;
  const float top = static_cast<float>(mapmax[blockIdx.y]) * scale;
  const float thr = static_cast<float>(0.1 * static_cast<double>(top));
  const int d = dist[static_cast<long>(blockIdx.y) * hw + i];
  label[static_cast<long>(blockIdx.y) * hw + i] = (d > 0 && static_cast<float>(d) * scale > thr) ? static_cast<int>(i) : -1;
}

// markers (heatmap.py:131-137): a core pixel carries its component's root + 2; a pixel further than two 3x3 dilations
// from the mask is background (1); the rest -- the blobs outside their cores and the ring around them -- is unknown (0).
// The frame of the map is a boundary (-1) from the start, as cv2.watershed makes it: it neither takes nor passes a label.
__global__ __launch_bounds__(kKpThreads) void kp_marker_kernel(int H, int W, const int* __restrict__ dist,
                                                               const int* __restrict__ label, int* __restrict__ marker) {
  const long hw = static_cast<long>(H) * W;
  const long i = blockIdx.x * static_cast<long>(kKpThreads) + threadIdx.x;
  if (i >= hw) return;
  const int* d = dist + static_cast<long>(blockIdx.y) * hw;
  const int l = label[static_cast<long>(blockIdx.y) * hw + i];
  const int y = static_cast<int>(i / W), x = static_cast<int>(i % W);
  int mk = l + 2;
  if (y == 0 || x == 0 || y == H - 1 || x == W - 1) {
    mk = -1;
  } else if (l < 0) {
    bool near = false;
    for (int dy = -2; dy <= 2; ++dy)
      for (int dx = -2; dx <= 2; ++dx) {
        const int yy = y + dy, xx = x + dx;
        if (yy >= 0 && yy < H && xx >= 0 && xx < W) near = near || d[static_cast<long>(yy) * W + xx] > 0;
      }
    mk = near ? 0 : 1;
  }
  marker[static_cast<long>(blockIdx.y) * hw + i] = mk;
}

// one round of the watershed of a two-valued image (cv2.watershed floods level by level; every difference between equal
// pixels is level 0): an unknown pixel with a labelled 4-neighbour of its own mask value takes the label its labelled
// neighbours agree on, or becomes a line (-1) where they differ.  Lines and the background marker do not count as a
// region later; lines are not propagated.  Rounds are synchronous (src -> dst): the result does not depend on timing.
__global__ __launch_bounds__(kKpThreads) void kp_flood_kernel(int H, int W, const int* __restrict__ dist,
                                                              const int* __restrict__ src, int* __restrict__ dst,
                                                              int* __restrict__ changed) {
  const long hw = static_cast<long>(H) * W;
  const long i = blockIdx.x * static_cast<long>(kKpThreads) + threadIdx.x;
  if (i >= hw) return;
  const int* d = dist + static_cast<long>(blockIdx.y) * hw;
  const int* in = src + static_cast<long>(blockIdx.y) * hw;
  int mk = in[i];
  if (mk == 0) {
    const int y = static_cast<int>(i / W), x = static_cast<int>(i % W);
    const bool inside = d[i] > 0;
    int lo = 0x7fffffff, hi = 0;
    bool active = false;
    const int ny[4] = {-1, 1, 0, 0}, nx[4] = {0, 0, -1, 1};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int yy = y + ny[k], xx = x + nx[k];
      if (yy < 0 || yy >= H || xx < 0 || xx >= W) continue;
      const long q = static_cast<long>(yy) * W + xx;
      const int l = in[q];
      if (l <= 0) continue;
      lo = min(lo, l);
      hi = max(hi, l);
      active = active || ((d[q] > 0) == inside);
    }
    if (active) {
      mk = lo == hi ? lo : -1;
      *changed = 1;
    }
  }
  dst[static_cast<long>(blockIdx.y) * hw + i] = mk;
}

// region labels for the peak / selection kernels: the root index of the core a pixel was flooded from, -1 elsewhere
__global__ __launch_bounds__(kKpThreads) void kp_region_kernel(int H, int W, const int* __restrict__ marker, int* __restrict__ label) {
  const long hw = static_cast<long>(H) * W;
  const long i = blockIdx.x * static_cast<long>(kKpThreads) + threadIdx.x;
  if (i >= hw) return;
  const int mk = marker[static_cast<long>(blockIdx.y) * hw + i];
  label[static_cast<long>(blockIdx.y) * hw + i] = mk >= 2 ? mk - 2 : -1;
}

// per region: (maximum of the thresholded map, first raster index attaining it), packed so that a 64-bit max does both
__global__ __launch_bounds__(kKpThreads) void kp_peak_kernel(const float* __restrict__ heat, int H, int W,
                                                             const float* __restrict__ thr_per_map,
                                                             const int* __restrict__ label, unsigned long long* __restrict__ best) {
  const long hw = static_cast<long>(H) * W;
  const long i = blockIdx.x * static_cast<long>(kKpThreads) + threadIdx.x;
  if (i >= hw) return;
  const int l = label[static_cast<long>(blockIdx.y) * hw + i];
  if (l < 0) return;
  float v = heat[static_cast<long>(blockIdx.y) * hw + i];
  if (v < thr_per_map[blockIdx.y]) v = 0.f;  // the median mask may cover zeroed pixels
  const unsigned long long key = (static_cast<unsigned long long>(__float_as_uint(v)) << 32) |
                                 static_cast<unsigned long long>(0xFFFFFFFFu - static_cast<unsigned>(i));
  atomicMax(&best[static_cast<long>(blockIdx.y) * hw + l], key);
}

__device__ __forceinline__ void kp_write_point(float* out, int rank, unsigned long long best_key, int W) {
  // (a region whose thresholded values are all zero: the reference's np.where(map == 0) then finds the first zero
  // of the WHOLE map, pixel 0 -- heatmap.py:168-170)
  const unsigned peak = static_cast<unsigned>(best_key >> 32);
  const unsigned idx = peak == 0u ? 0u : 0xFFFFFFFFu - static_cast<unsigned>(best_key & 0xFFFFFFFFull);
  out[2 * rank] = static_cast<float>(idx % static_cast<unsigned>(W));      // x
  out[2 * rank + 1] = static_cast<float>(idx / static_cast<unsigned>(W));  // y
}

// one workgroup per map: rank the roots by (peak descending, root index ascending), write the first `num`.
// Up to max_regions roots are gathered into `cand` and ranked against each other; a map with MORE regions (a speckled
// map of an untrained net can have tens of thousands) takes the exact selection over all roots instead: `num` rounds
// of a workgroup-wide arg-max of the 64-bit key (peak bits, ~root) below the key selected in the round before --
// the same order, no cap, no dependence on the order the atomics arrived in (the reference's extract_points_ has no
// region limit either, heatmap.py:148-200).
__global__ __launch_bounds__(1024) void kp_select_kernel(int H, int W, const int* __restrict__ label,
                                                         const unsigned long long* __restrict__ best, int num,
                                                         int max_regions, unsigned long long* __restrict__ cand,
                                                         float* __restrict__ points, int* __restrict__ counts) {
  __shared__ int n_cand;
  __shared__ unsigned long long red[1024 / 64];
  __shared__ unsigned long long chosen;
  const long hw = static_cast<long>(H) * W;
  (void)label;
  const unsigned long long* bst = best + static_cast<long>(blockIdx.x) * hw;
  unsigned long long* cd = cand + static_cast<long>(blockIdx.x) * max_regions * 2;  // (peak key, root)
  if (threadIdx.x == 0) n_cand = 0;
  __syncthreads();
  // a root = the first pixel of a region that owns at least one pixel: every labelled pixel has put a non-zero key into
  // best[its label] (kp_peak_kernel).  (Not `lab[i] == i`: a watershed region whose core starts in the frame of the map
  // keeps its label, but the frame pixel itself belongs to no region.)
  for (long i = threadIdx.x; i < hw; i += blockDim.x) {
    if (bst[i] != 0ull) {
      const int slot = atomicAdd(&n_cand, 1);
      if (slot < max_regions) {
        cd[2 * slot] = bst[i];
        cd[2 * slot + 1] = static_cast<unsigned long long>(i);
      }
    }
  }
  __syncthreads();
  const int total = n_cand;
  float* out = points + static_cast<long>(blockIdx.x) * num * 2;
  for (int k = threadIdx.x; k < num * 2; k += blockDim.x) out[k] = -1.f;
  __syncthreads();
  if (threadIdx.x == 0) counts[blockIdx.x] = total;  // regions found
  if (total <= max_regions) {
    for (int c = threadIdx.x; c < total; c += blockDim.x) {
      const unsigned peak = static_cast<unsigned>(cd[2 * c] >> 32);
      const unsigned long long root = cd[2 * c + 1];
      int rank = 0;
      for (int o = 0; o < total; ++o) {
        const unsigned po = static_cast<unsigned>(cd[2 * o] >> 32);
        if (po > peak || (po == peak && cd[2 * o + 1] < root)) ++rank;
      }
      if (rank < num) kp_write_point(out, rank, cd[2 * c], W);
    }
    return;
  }
  // more roots than the candidate buffer holds: exact top-`num` over ALL roots (uniform branch: `total` is shared)
  unsigned long long below = ~0ull;  // keys are < 2^64 - 1: a root index is < 2^31
  for (int r = 0; r < num; ++r) {
    unsigned long long mine = 0ull;  // 0 = nothing (a real key has non-zero low half: ~root with root < 2^31)
    for (long i = threadIdx.x; i < hw; i += blockDim.x) {
      if (bst[i] == 0ull) continue;
      const unsigned long long key = (bst[i] & 0xFFFFFFFF00000000ull) | (0xFFFFFFFFull - static_cast<unsigned long long>(i));
      if (key < below && key > mine) mine = key;
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
      const unsigned long long o = __shfl_xor(mine, off);
      mine = o > mine ? o : mine;
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
      unsigned long long m = 0ull;
      for (int k = 0; k < 1024 / 64; ++k) m = red[k] > m ? red[k] : m;
      chosen = m;
      if (m != 0ull) kp_write_point(out, r, bst[0xFFFFFFFFull - (m & 0xFFFFFFFFull)], W);
    }
    __syncthreads();
    below = chosen;
    if (below == 0ull) break;  // fewer than num regions (cannot happen on this branch; keeps the loop bounded)
  }
}

}  // namespace
}  // namespace unetpp

using namespace unetpp;

extern "C" int64_t unetpp_heatmap_pattern_workspace_bytes(int32_t N, int32_t n_maps, int32_t H, int32_t W) {
  if (N < 1 || n_maps < 1 || H < 1 || W < 1) return 0;
  const int64_t blocks = (static_cast<int64_t>(H) * W + kKpThreads - 1) / kKpThreads;
  return static_cast<int64_t>(N) * n_maps * blocks * 4;
}

extern "C" int unetpp_heatmap_pattern(const float* points, int32_t N, int32_t P, const int32_t* map_points,
                                      const int32_t* map_begin, int32_t n_maps, int32_t H, int32_t W, float radius,
                                      float* out_nchw, void* workspace, void* stream) {
  if (!points || !map_points || !map_begin || !out_nchw || !workspace) return UNETPP_EINVAL;
  if (N < 1 || P < 1 || n_maps < 1 || H < 1 || W < 1 || !(radius > 0.f) || static_cast<long>(N) * n_maps > 65535)
    return UNETPP_EINVAL;
  const unsigned blocks = static_cast<unsigned>((static_cast<long>(H) * W + kKpThreads - 1) / kKpThreads);
  hipStream_t st = static_cast<hipStream_t>(stream);
  float* blockmax = static_cast<float*>(workspace);
  hipLaunchKernelGGL(pattern_map_kernel, dim3(blocks, N * n_maps), dim3(kKpThreads), 0, st, points, P, map_points, map_begin,
                     n_maps, H, W, static_cast<double>(radius), out_nchw, blockmax);
  hipLaunchKernelGGL(pattern_norm_kernel, dim3(blocks, N * n_maps), dim3(kKpThreads), 0, st, H, W, out_nchw, blockmax);
  return launch_status();
}

extern "C" int64_t unetpp_keypoints_workspace_bytes(int32_t maps, int32_t H, int32_t W, int32_t max_regions) {
  if (maps < 1 || H < 1 || W < 1 || max_regions < 1) return 0;
  const int64_t hw = static_cast<int64_t>(H) * W;
  return static_cast<int64_t>(maps) * (hw * 4 * 3 + hw * 8 + static_cast<int64_t>(max_regions) * 16 + 4) + 64;
}

// stage 0: mask + labels initialised; stage 1: `sweeps` merge + flatten sweeps (sets *changed when a sweep moved a
// label: the caller repeats stage 1 until it stays 0); stage 2: peaks + selection.  heat [maps, H, W]; thr [maps];
// points [maps, num, 2] as (x, y), -1 where there is no region; counts [maps] = regions found.
// The reference's region step sits between 0 and 2: 3 = distance initialised from the mask; 4 = `sweeps` relaxation
// sweeps (repeat until *changed stays 0); 5 = cores -> labels (then stage 1 until stable: components of the cores);
// 6 = markers; 7 = `sweeps` pairs of flood rounds (repeat until *changed stays 0); 8 = region labels; then stage 2.
extern "C" int unetpp_keypoints_extract(int32_t stage, const float* heat, int32_t maps, int32_t H, int32_t W,
                                        const float* thr_per_map, int32_t num, int32_t max_regions, int32_t sweeps,
                                        void* workspace, int32_t* changed, float* points, int32_t* counts, void* stream) {
  if (!heat || !thr_per_map || !workspace || maps < 1 || maps > 65535 || H < 1 || W < 1 || num < 1 || max_regions < num ||
      static_cast<long>(H) * W >= 0x7fffffffL)
    return UNETPP_EINVAL;
  const long hw = static_cast<long>(H) * W;
  unsigned long long* best = static_cast<unsigned long long*>(workspace);  // 8-byte aligned first
  unsigned long long* cand = best + static_cast<long>(maps) * hw;
  int* label = reinterpret_cast<int*>(cand + static_cast<long>(maps) * max_regions * 2);
  int* dist = label + static_cast<long>(maps) * hw;
  int* marker = dist + static_cast<long>(maps) * hw;
  int* mapmax = marker + static_cast<long>(maps) * hw;
  const dim3 grid(static_cast<unsigned>((hw + kKpThreads - 1) / kKpThreads), static_cast<unsigned>(maps));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (stage == 0) {
    hipLaunchKernelGGL(kp_mask_kernel, grid, dim3(kKpThreads), 0, st, heat, H, W, thr_per_map, label, best);
  } else if (stage == 1) {
    if (!changed || sweeps < 1) return UNETPP_EINVAL;
    for (int s = 0; s < sweeps; ++s) {
      hipLaunchKernelGGL(kp_merge_kernel, grid, dim3(kKpThreads), 0, st, H, W, label, changed);
      hipLaunchKernelGGL(kp_flatten_kernel, grid, dim3(kKpThreads), 0, st, H, W, label);
    }
  } else if (stage == 2) {
    if (!points || !counts) return UNETPP_EINVAL;
    hipLaunchKernelGGL(kp_peak_kernel, grid, dim3(kKpThreads), 0, st, heat, H, W, thr_per_map, label, best);
    hipLaunchKernelGGL(kp_select_kernel, dim3(static_cast<unsigned>(maps)), dim3(1024), 0, st, H, W, label, best, num,
                       max_regions, cand, points, counts);
  } else if (stage == 3) {
    hipLaunchKernelGGL(kp_dist_init_kernel, grid, dim3(kKpThreads), 0, st, H, W, label, dist, mapmax);
  } else if (stage == 4) {
    if (!changed || sweeps < 1) return UNETPP_EINVAL;
    for (int s = 0; s < sweeps; ++s)
      hipLaunchKernelGGL(kp_dist_sweep_kernel, grid, dim3(kKpThreads), 0, st, H, W, dist, changed);
  } else if (stage == 5) {
    hipLaunchKernelGGL(kp_dist_max_kernel, grid, dim3(kKpThreads), 0, st, H, W, dist, mapmax);
    hipLaunchKernelGGL(kp_core_kernel, grid, dim3(kKpThreads), 0, st, H, W, dist, mapmax, label);
  } else if (stage == 6) {
    hipLaunchKernelGGL(kp_marker_kernel, grid, dim3(kKpThreads), 0, st, H, W, dist, label, marker);
  } else if (stage == 7) {
    if (!changed || sweeps < 1) return UNETPP_EINVAL;
    for (int s = 0; s < sweeps; ++s) {  // pairs: the current state is always in `marker` between calls
      hipLaunchKernelGGL(kp_flood_kernel, grid, dim3(kKpThreads), 0, st, H, W, dist, marker, label, changed);
      hipLaunchKernelGGL(kp_flood_kernel, grid, dim3(kKpThreads), 0, st, H, W, dist, label, marker, changed);
    }
  } else if (stage == 8) {
    hipLaunchKernelGGL(kp_region_kernel, grid, dim3(kKpThreads), 0, st, H, W, marker, label);
  } else {
    return UNETPP_EINVAL;
  }
  return launch_status();
}

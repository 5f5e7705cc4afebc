// Pointwise (taps = 1) fp32 weight gradient with BOTH operands loaded straight into the MFMA operand registers -- the
// weight gradient of the 2x2 stride-2 transposed convolution (x: one view; dy: the four pixel phases of d_up) and of the
// 1x1 convolution (autograd of /root/reference/models/unet.py:187,191):
//
//   dW[k][n] = sum over pixels p of x[p][k] * dy[p][n],   db[n] = sum over p of dy[p][n]
//
// wgrad_dma_kernel<1> gives every (32-channel, 32-column) pair of the product a workgroup of its own, so x is staged
// (through LDS) once per column tile and dy once per channel tile: 1 072 MB through the CUs for the 402 MB of a level-0
// layer, at 2.9 TB/s.  Here the contraction index of v_mfma_f32_16x16x4_f32 is the PIXEL (four per instruction):
//   * lane (t16, g) loads 16 bytes = channels 4 t16 .. + 3 of pixel p0 + g of a 64-channel block of x (one instruction
//     = four whole 256-byte runs) and the same of two 64-column halves of a 128-column block of dy; MFMA (m, q) of a
//     step takes element m of the x load as its A operand (rows = channels 4 i + m) and element q of a dy load as B
//     (columns 4 j + q): 32 MFMAs per three loads, no LDS, no barrier in the loop, three steps in flight;
//   * a wave keeps a whole 64 x 128 block of dW in 128 accumulator registers; the eight waves of a workgroup take eight
//     pixel ranges of ONE block and are summed through LDS in a fixed order (bitwise reproducible), so a slab is
//     written by (blocks of dW) workgroups and there are 256 / blocks slabs: 8.4 MB of slabs at every level of the
//     network (one slab per workgroup over all blocks would be 67 MB at level 2: K 256, N 512);
//   * every operand byte comes from HBM once: at level 0 (K 64, N 128) dW is one block; at the deeper levels the
//     workgroups of a pixel range sit next to each other on one XCD, and the re-reads of the other blocks hit its L2.
// Measured on the level-0 layer of BASELINE configs[1]: 87 us against 140 us (tools/probes/pw_wgrad_probe.hip).
#include "common.h"

namespace unetpp {
namespace {

constexpr int kPwThreads = 512;
constexpr int kDepth = 3;  // steps (4 pixels, three 16-byte loads per lane) in flight

struct WPwArgs {
  unetpp_wgrad_desc d;
  int K, N;
  int kb_count, nb_count;  // 64-channel / 128-column blocks of dW
  long rows;               // N_img * H image rows
};

// the (view, channel) a global channel / column index falls into; views of one operand share their tensor geometry
__device__ __forceinline__ const float* lane_base(const unetpp_view* v, int n_views, int c) {
  int i = 0;
  while (i < n_views - 1 && c >= v[i].c_len) {
    c -= v[i].c_len;
    ++i;
  }
  return v[i].ptr + (static_cast<long>(v[i].oy) * v[i].Ws + v[i].ox) * v[i].C + v[i].c_off + c;
}

__global__ __launch_bounds__(kPwThreads, 2) void wgrad_pw_kernel(const WPwArgs a) {
  extern __shared__ __attribute__((aligned(16))) float red_lds[];  // the reduction's regions: [4][32 accumulators][64 lanes][4]
  const unetpp_wgrad_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, t16 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  constexpr int n_ps = 8;  // pixel ranges (waves) per workgroup
  const int ps = wave;
  // workgroup -> (slab = pixel-range group, block of dW): blocks b and b + 8 share an XCD (round-robin dispatch), so the
  // workgroups of one XCD are numbered consecutively and the blocks of a slab are neighbours there
  const int blocks = a.kb_count * a.nb_count, n_wg = gridDim.x;
  int wg = blockIdx.x;
  if ((n_wg & 7) == 0) wg = (blockIdx.x & 7) * (n_wg >> 3) + (blockIdx.x >> 3);
  const int split = wg / blocks, sb = wg - split * blocks;
  const int kb = sb / a.nb_count, nb = sb - kb * a.nb_count;

  // per-lane operand bases (the views of an operand share C, Ws, sy, sx: launcher) and pixel strides
  const float* xa = lane_base(d.x, d.n_x, 64 * kb + 4 * t16);
  const float* yb0 = lane_base(d.dy, d.n_dy, 128 * nb + 4 * t16);
  const float* yb1 = lane_base(d.dy, d.n_dy, 128 * nb + 64 + 4 * t16);
  const unetpp_view& X0 = d.x[0];
  const unetpp_view& Y0 = d.dy[0];
  const long x_row = static_cast<long>(X0.sy) * X0.Ws * X0.C, y_row = static_cast<long>(Y0.sy) * Y0.Ws * Y0.C;
  const int x_px = X0.sx * X0.C, y_px = Y0.sx * Y0.C;

  // image rows of this wave: the workgroup's share of all rows, split over the block's pixel ranges
  const long parts = static_cast<long>(d.n_split) * n_ps, part = static_cast<long>(split) * n_ps + ps;
  const long r0 = a.rows * part / parts, r1 = a.rows * (part + 1) / parts;
  const int steps_per_row = d.W >> 2;
  const long n_steps = (r1 - r0) * steps_per_row;

  f32x4 acc[4][2][4];
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[m][h][q] = f32x4{0.f, 0.f, 0.f, 0.f};
  f32x4 dbs[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};

  f32x4 A[kDepth], B[kDepth][2];
  long i_row = r0;  // cursor of the next step to request
  int i_x = 0;
  auto issue = [&](int slot) {
    const long xo = i_row * x_row + static_cast<long>(i_x + g) * x_px, yo = i_row * y_row + static_cast<long>(i_x + g) * y_px;
    A[slot] = *reinterpret_cast<const f32x4*>(xa + xo);
    B[slot][0] = *reinterpret_cast<const f32x4*>(yb0 + yo);
    B[slot][1] = *reinterpret_cast<const f32x4*>(yb1 + yo);
    i_x += 4;
    if (i_x == d.W) {
      i_x = 0;
      ++i_row;
    }
  };
#pragma unroll
  for (int i = 0; i < kDepth; ++i)
    if (i < n_steps) issue(i);
  for (long s = 0; s < n_steps; s += kDepth) {
#pragma unroll
    for (int slot = 0; slot < kDepth; ++slot) {
      if (s + slot < n_steps) {  // uniform
        const f32x4 av = A[slot], b0 = B[slot][0], b1 = B[slot][1];
        if (s + slot + kDepth < n_steps) issue(slot);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          dbs[0][e] += b0[e];
          dbs[1][e] += b1[e];
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            acc[m][0][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], b0[q], acc[m][0][q], 0, 0, 0);
            acc[m][1][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[m], b1[q], acc[m][1][q], 0, 0, 0);
          }
      }
    }
  }

  // ---- the pixel ranges of a block, summed in a fixed order: in round `half` = n_ps / 2, n_ps / 4, .. 1 the ranges
  // half .. 2 half - 1 write their accumulators (32 KB per wave, lane linear) and the ranges 0 .. half - 1 add them ----
  for (int half = n_ps >> 1; half >= 1; half >>= 1) {
    __syncthreads();  // the regions are free (previous round read)
    if (ps >= half && ps < 2 * half) {
      float* rg = red_lds + (ps - half) * (34 * 256) + lane * 4;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(rg + ((m * 2 + h) * 4 + q) * 256) = acc[m][h][q];
      *reinterpret_cast<f32x4*>(rg + 32 * 256) = dbs[0];
      *reinterpret_cast<f32x4*>(rg + 33 * 256) = dbs[1];
    }
    __syncthreads();
    if (ps < half) {
      const float* rg = red_lds + ps * (34 * 256) + lane * 4;
#pragma unroll
      for (int m = 0; m < 4; ++m)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int q = 0; q < 4; ++q) acc[m][h][q] += *reinterpret_cast<const f32x4*>(rg + ((m * 2 + h) * 4 + q) * 256);
      dbs[0] += *reinterpret_cast<const f32x4*>(rg + 32 * 256);
      dbs[1] += *reinterpret_cast<const f32x4*>(rg + 33 * 256);
    }
  }
  if (ps != 0) return;

  // ---- slab of this workgroup: [K][N] then the db row.  Register rr of acc[m][h][q] of lane (t16, g) is
  // dW[64 kb + 4 (4 g + rr) + m][128 nb + 64 h + 4 t16 + q]: the four q of a lane are one 16-byte piece ----
  float* slab = d.slabs + static_cast<long>(split) * (static_cast<long>(a.K) + 1) * a.N;
#pragma unroll
  for (int m = 0; m < 4; ++m)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const int k = 64 * kb + 4 * (4 * g + rr) + m, n = 128 * nb + 64 * h + 4 * t16;
        *reinterpret_cast<f32x4*>(slab + static_cast<long>(k) * a.N + n) =
            f32x4{acc[m][h][0][rr], acc[m][h][1][rr], acc[m][h][2][rr], acc[m][h][3][rr]};
      }
  if (kb == 0) {  // db: the four pixel slots of a step (g) are the lanes 16 and 32 apart
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f32x4 v = dbs[h];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] += __shfl_xor(v[e], 16);
        v[e] += __shfl_xor(v[e], 32);
      }
      if (g == 0) *reinterpret_cast<f32x4*>(slab + static_cast<long>(a.K) * a.N + 128 * nb + 64 * h + 4 * t16) = v;
    }
  }
}

bool same_geometry(const unetpp_view* v, int n) {
  for (int i = 1; i < n; ++i)
    if (v[i].C != v[0].C || v[i].Hs != v[0].Hs || v[i].Ws != v[0].Ws || v[i].sy != v[0].sy || v[i].sx != v[0].sx) return false;
  return true;
}

bool pw_plain(const unetpp_view& v) {
  return v.scale == nullptr && v.shift == nullptr && v.gate == nullptr && !v.relu && ((v.C | v.c_off | v.c_len) & 3) == 0 &&
         (reinterpret_cast<uintptr_t>(v.ptr) & 15) == 0;
}

// the block structure of a launch this kernel takes; false when another kernel has to
bool wgrad_pw_shape(const unetpp_wgrad_desc* d, WPwArgs& a) {
  if (d == nullptr || d->taps != 1 || (d->flags & UNETPP_GEMM_BF16) != 0 || opt_value(OPT_PW_DIRECT, 1) == 0) return false;
  if ((d->W & 3) != 0 || d->n_x < 1 || d->n_dy < 1 || d->n_x > UNETPP_MAX_VIEWS || d->n_dy > UNETPP_MAX_VIEWS) return false;
  a.K = a.N = 0;
  for (int i = 0; i < d->n_x; ++i) {
    if (!view_ok(d->x[i]) || !pw_plain(d->x[i])) return false;
    a.K += d->x[i].c_len;
  }
  for (int i = 0; i < d->n_dy; ++i) {
    if (!view_ok(d->dy[i]) || !pw_plain(d->dy[i])) return false;
    a.N += d->dy[i].c_len;
  }
  if ((a.K & 63) != 0 || (a.N & 127) != 0) return false;
  if (!same_geometry(d->x, d->n_x) || !same_geometry(d->dy, d->n_dy)) return false;
  // a 16-byte piece never straddles two views; image rows follow each other at the row stride (no vertical padding)
  for (int i = 0; i < d->n_x; ++i)
    if (d->x[i].Hs != d->x[i].sy * d->H) return false;
  for (int i = 0; i < d->n_dy; ++i)
    if (d->dy[i].Hs != d->dy[i].sy * d->H) return false;
  a.kb_count = a.K >> 6;
  a.nb_count = a.N >> 7;
  if (static_cast<long>(a.kb_count) * a.nb_count * 4096 > 0x7fffffffL) return false;  // (n_split <= 4096 workgroups per block)
  a.rows = static_cast<long>(d->N) * d->H;
  return true;
}

}  // namespace

// (32 x 32) pairs of dW a workgroup covers, 0 when the kernel does not take the descriptor (unetpp_wgrad_pairs_per_workgroup)
int wgrad_pw_pairs(const unetpp_wgrad_desc* d) {
  WPwArgs a;
  return wgrad_pw_shape(d, a) ? 8 : 0;  // one 64 x 128 block
}

// returns UNETPP_OK after launching, or 1 when the descriptor needs another kernel
int launch_wgrad_pw(const unetpp_wgrad_desc* d, hipStream_t st) {
  WPwArgs a;
  if (!wgrad_pw_shape(d, a)) return 1;
  a.d = *d;
  static bool raised[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return UNETPP_ELAUNCH;
  constexpr size_t lds_bytes = 4 * 34 * 256 * sizeof(float);  // four regions of 34 KB
  if (!raised[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_pw_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            static_cast<int>(lds_bytes)) != hipSuccess)
      return UNETPP_ELAUNCH;
    raised[dev] = true;
  }
  const dim3 grid(static_cast<unsigned>(d->n_split) * static_cast<unsigned>(a.kb_count * a.nb_count));
  hipLaunchKernelGGL(wgrad_pw_kernel, grid, dim3(kPwThreads), lds_bytes, st, a);
  note_kernel("wgrad_pw_kernel");
  return launch_status();
}

}  // namespace unetpp

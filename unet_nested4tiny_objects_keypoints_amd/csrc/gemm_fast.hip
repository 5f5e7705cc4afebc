// Fast path of the multi-view pixel GEMM (same math and MFMA maps as gemm_pix.hip) for plain, 16-byte
// aligned input views -- every 3x3 / deconv / 1x1 launch of the network except the 1- or 3-channel first layer.
//
// What changes against the generic kernel:
//   * software pipeline: the global loads of K-chunk c+1 (input patch + weight image) are issued into
//     registers BEFORE the MFMA loop of chunk c and written to LDS after it, so HBM/L2 latency hides
//     under ~9 k cycles of MFMA work per chunk instead of stalling the workgroup;
//   * the weights arrive as a pre-built LDS image (unetpp_gemm_pack_weight_image): staging them is a
//     straight 16-byte copy, no strided gathers, no index math;
//   * LDS: the input patch keeps the padded 80-byte pixel stride (conflict-free ds_read_b128 AND every tap's
//     address is the lane's base plus a compile-time offset -- an XOR swizzle here costs 36 address registers),
//     the weight image is XOR-swizzled instead of padded (32-B columns, half ^= (col>>3)&1): 45.6 KB instead
//     of 55 KB per workgroup, so three workgroups fit a CU;
//   * the per-item pixel geometry (halo position, bounds) is computed once per workgroup, per view only
//     the base offset is refreshed;
//   * one linear grid with the 32-column tile as the fastest index (consecutive workgroups re-read the
//     same input patch from L2) and an XCD-aware bijective remap, so neighbouring patches share an L2.
#include "common.h"

namespace unetpp {
namespace {

constexpr int KC = 16;

struct FastArgs {
  unetpp_gemm_desc d;
  int log2tw, tiles_x, tiles_y;
  int Ktot, Ncols, n_tiles, n_chunks;
  long total_blocks;
};

__device__ __forceinline__ long xcd_remap(long bid, long total) {
  const long q = total >> 3, r = total & 7;
  const long xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// image of one (column tile, K chunk): [tap][g 2][col 32][half' 2][4] floats, half' = half ^ ((col>>3)&1)
template <int TAPS>
__global__ void pack_image_kernel(const FastArgs a, float* __restrict__ img) {
  constexpr int IMG = TAPS * 512;
  const long total = static_cast<long>(a.n_tiles) * a.n_chunks * IMG;
  const long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
  if (i >= total) return;
  const int e = i & 3, hs = (i >> 2) & 1, j = (i >> 3) & 31, g = (i >> 8) & 1;
  long r = i >> 9;
  const int tap = static_cast<int>(r % TAPS);
  r /= TAPS;
  int chunk = static_cast<int>(r % a.n_chunks);
  int nt = static_cast<int>(r / a.n_chunks);
  const int h = hs ^ ((j >> 3) & 1);
  const int kk = 8 * g + 4 * h + e;
  // chunk -> (view, first channel)
  int kbase = 0, s = 0;
  for (; s < a.d.n_in; ++s) {
    const int ch = (a.d.in[s].c_len + KC - 1) / KC;
    if (chunk < ch) break;
    chunk -= ch;
    kbase += a.d.in[s].c_len;
  }
  const int kin = chunk * KC + kk;
  const bool k_ok = kin < a.d.in[s].c_len;
  // column tile -> (out view, first column)
  int col_base = 0, ov = 0;
  for (; ov < a.d.n_out; ++ov) {
    const int tv = (a.d.out[ov].c_len + 31) >> 5;
    if (nt < tv) break;
    nt -= tv;
    col_base += a.d.out[ov].c_len;
  }
  const int cin = nt * 32 + j;
  const bool n_ok = cin < a.d.out[ov].c_len;
  float v = 0.f;
  if (k_ok && n_ok) v = a.d.weight[(static_cast<long>(tap) * a.Ktot + kbase + kin) * a.Ncols + col_base + cin];
  img[i] = v;
}

template <int TAPS>
__global__ __launch_bounds__(kThreads, 3) void gemm_fast_kernel(const FastArgs a) {
  constexpr int HALO = (TAPS == 9) ? 1 : 0;
  constexpr int MAXPIX = (TAPS == 9) ? kMaxHaloPixels : kBlockPixels;
#ifdef UNETPP_A_SWIZZLE
  constexpr int KCP = 16;
#define A_ADDR(hp, slot) ((hp) * 16 + ((((slot)) ^ (((hp) >> 2) & 3)) << 2))
#else
  constexpr int KCP = 20;  // input pixel stride in LDS (floats)
#define A_ADDR(hp, slot) ((hp) * 20 + ((slot) << 2))
#endif
  constexpr int IN_FLOATS = MAXPIX * KCP;
  constexpr int IMG = TAPS * 512;
  constexpr int IN_ITEMS = (MAXPIX * 4 + kThreads - 1) / kThreads;
  constexpr int W_ITEMS = (IMG / 4 + kThreads - 1) / kThreads;
  __shared__ __attribute__((aligned(16))) float smem[IN_FLOATS + IMG];
  float* in_tile = smem;
  float* w_tile = smem + IN_FLOATS;

  const unetpp_gemm_desc& d = a.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;

  // ---- which pixel patch, which 32 output columns (column tile fastest, XCD-contiguous) ----
  const long lb = xcd_remap(blockIdx.x, a.total_blocks);
  int nt = static_cast<int>(lb % a.n_tiles);
  const int nt_global = nt;
  long bid = lb / a.n_tiles;
  const int txi = static_cast<int>(bid % a.tiles_x);
  bid /= a.tiles_x;
  const int tyi = static_cast<int>(bid % a.tiles_y);
  const int n = static_cast<int>(bid / a.tiles_y);
  const int TW = 1 << a.log2tw, TH = kBlockPixels >> a.log2tw;
  const int ty0 = tyi * TH, tx0 = txi * TW;
  const int HWp = TW + 2 * HALO, HHp = TH + 2 * HALO;
  const int npix = HWp * HHp;

  int ov = 0, col_base = 0;
  while (ov < d.n_out - 1) {
    const int tiles_v = (d.out[ov].c_len + 31) >> 5;
    if (nt < tiles_v) break;
    nt -= tiles_v;
    col_base += d.out[ov].c_len;
    ++ov;
  }
  const unetpp_view& O = d.out[ov];
  const int n0 = col_base + nt * 32;
  const int n_cnt = min(32, O.c_len - nt * 32);

  // ---- per-thread staging items.  Kept in registers across the K loop: one 32-bit element offset per item and
  // one bit per item (pixel inside the image); everything else is recomputed from tid where it is needed, so
  // the MFMA loop keeps enough registers to software-pipeline its LDS reads. ----
  unsigned in_mask = 0;
#pragma unroll
  for (int q = 0; q < IN_ITEMS; ++q) {
    const int it = tid + q * kThreads;
    const int hp = it >> 2;
    const int hy = hp / HWp, hx = hp - hy * HWp;
    const int y = ty0 + hy - HALO, x = tx0 + hx - HALO;
    if ((it < npix * 4) && y >= 0 && y < d.H && x >= 0 && x < d.W) in_mask |= 1u << q;
  }
  int apix[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int p = 64 * wave + 32 * mt + j;
    apix[mt] = (p >> a.log2tw) * HWp + (p & (TW - 1));
  }
  const int wb = j * 8 + ((h ^ ((j >> 3) & 1)) << 2);

  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  f32x4 reg_in[IN_ITEMS], reg_w[W_ITEMS];
  // element offsets, unsigned 32-bit so the loads use the scalar-base + vector-offset form (no 64-bit address
  // register pairs); the launcher only takes this path for tensors below 2^31 elements
  unsigned voff[IN_ITEMS];
  int pf_cnt = 0;      // valid channels of the chunk currently held in reg_in
  const float* wimg = d.weight_image + static_cast<long>(nt_global) * a.n_chunks * IMG;

  // chunk cursor
  int s = 0, c0 = 0;
  // The prefetch is straight-line code: every item loads from a VALID address (out-of-image halo pixels and
  // padding items are clamped into the image, channels past the view to channel 0) and the zeroing happens
  // at the LDS write.  Conditional loads would put the loads under divergent branches, where hipcc drains
  // vmcnt at every join and the prefetch stops overlapping the MFMA loop.
  auto view_offsets = [&](const unetpp_view& V) {
#pragma unroll
    for (int q = 0; q < IN_ITEMS; ++q) {
      const int hp = (tid + q * kThreads) >> 2;
      const int hy = hp / HWp, hx = hp - hy * HWp;
      const int yy = min(max(ty0 + hy - HALO, 0), d.H - 1), xx = min(max(tx0 + hx - HALO, 0), d.W - 1);
      voff[q] = static_cast<unsigned>(view_pixel_offset(V, n, yy, xx));
    }
  };
  auto load_chunk = [&](const unetpp_view& V, int cbeg, int chunk_idx) {
    pf_cnt = min(KC, V.c_len - cbeg);
#pragma unroll
    for (int q = 0; q < IN_ITEMS; ++q) {
      const int cc = ((tid + q * kThreads) & 3) << 2;
      const unsigned off = voff[q] + static_cast<unsigned>(cbeg + (cc < pf_cnt ? cc : 0));
      reg_in[q] = *reinterpret_cast<const f32x4*>(V.ptr + off);
    }
    const float* wp = wimg + static_cast<long>(chunk_idx) * IMG;
#pragma unroll
    for (int q = 0; q < W_ITEMS; ++q) {
      const unsigned it = min(tid + q * kThreads, IMG / 4 - 1);
      reg_w[q] = *reinterpret_cast<const f32x4*>(wp + it * 4u);
    }
  };
  auto store_chunk = [&]() {
#pragma unroll
    for (int q = 0; q < IN_ITEMS; ++q) {
      const int it = tid + q * kThreads;
      const int hp = it >> 2, q4 = it & 3;
      const bool keep = ((in_mask >> q) & 1u) && (q4 << 2) < pf_cnt;
      f32x4 v = reg_in[q];
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = keep ? v[e] : 0.f;
      if (it < npix * 4) *reinterpret_cast<f32x4*>(&in_tile[A_ADDR(hp, q4)]) = v;
    }
#pragma unroll
    for (int q = 0; q < W_ITEMS; ++q) {
      const int it = tid + q * kThreads;
      if (it < IMG / 4) *reinterpret_cast<f32x4*>(&w_tile[it * 4]) = reg_w[q];
    }
  };
  // one (tap, 8-channel group) step of LDS fragments: B for the 32 columns, A for both pixel tiles
  struct Frag {
    f32x4 b, a0, a1;
  };
  auto read_frag = [&](int step) {
    const int tap = step >> 1, g = step & 1;
    const int tpix = (TAPS == 9) ? (tap / 3) * HWp + (tap % 3) : 0;
    Frag f;
    f.b = *reinterpret_cast<const f32x4*>(&w_tile[step * 256 + wb]);
    const int hp0 = apix[0] + tpix, hp1 = apix[1] + tpix;
    f.a0 = *reinterpret_cast<const f32x4*>(&in_tile[A_ADDR(hp0, 2 * g + h)]);
    f.a1 = *reinterpret_cast<const f32x4*>(&in_tile[A_ADDR(hp1, 2 * g + h)]);
    return f;
  };

  view_offsets(d.in[0]);
  load_chunk(d.in[0], 0, 0);
  store_chunk();
  __syncthreads();

  for (int chunk = 0; chunk < a.n_chunks; ++chunk) {
    // advance the cursor and prefetch the next chunk into registers
    int s2 = s, c2 = c0 + KC;
    if (c2 >= d.in[s].c_len) {
      ++s2;
      c2 = 0;
    }
    const bool more = chunk + 1 < a.n_chunks;
    if (more) {
      if (s2 != s) view_offsets(d.in[s2]);
      load_chunk(d.in[s2], c2, chunk + 1);
    }
    // ---- LDS -> MFMA for the current chunk: the fragments of step k+1 are read while the 8 MFMAs of step k
    // issue (both 8-channel groups always run; a short last chunk is zero-padded in LDS) ----
    Frag cur = read_frag(0);
#pragma unroll
    for (int step = 0; step < TAPS * 2; ++step) {
      Frag nxt = cur;
      if (step + 1 < TAPS * 2) nxt = read_frag(step + 1);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a0[t], cur.b[t], acc[0], 0, 0, 0);
#pragma unroll
      for (int t = 0; t < 4; ++t) acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a1[t], cur.b[t], acc[1], 0, 0, 0);
      cur = nxt;
    }
    __syncthreads();
    if (more) {
      store_chunk();
      __syncthreads();
    }
    s = s2;
    c0 = c2;
  }

  // ---- epilogue: bias, ReLU, gate, store / accumulate, optional BatchNorm partial sums ----
  const bool col_ok = j < n_cnt;
  const float bj = (d.bias != nullptr && col_ok) ? d.bias[n0 + j] : 0.f;
  float s1 = 0.f, s2sum = 0.f;
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = (r & 3) + 8 * (r >> 2) + 4 * h;
      const int p = 64 * wave + 32 * mt + i;
      const int y = ty0 + (p >> a.log2tw), x = tx0 + (p & (TW - 1));
      if (col_ok && y < d.H && x < d.W) {
        float v = acc[mt][r] + bj;
        if (O.relu) v = fmaxf(v, 0.f);
        s1 += v;
        s2sum += v * v;
        const long off = view_pixel_offset(O, n, y, x) + nt * 32 + j;
        if (O.gate != nullptr && !O.gate_sum) v = (O.gate[off] > 0.f) ? v : 0.f;
        if (O.accumulate) v += O.ptr[off];
        if (O.gate != nullptr && O.gate_sum) v = (O.gate[off] > 0.f) ? v : 0.f;
        O.ptr[off] = v;
      }
    }
  }
  if (d.stats_partial != nullptr) {
    s1 += __shfl_xor(s1, 32);
    s2sum += __shfl_xor(s2sum, 32);
    if (h == 0) {
      smem[(wave * 32 + j) * 2 + 0] = s1;
      smem[(wave * 32 + j) * 2 + 1] = s2sum;
    }
    __syncthreads();
    if (tid < n_cnt) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        t1 += smem[(w * 32 + tid) * 2 + 0];
        t2 += smem[(w * 32 + tid) * 2 + 1];
      }
      // partial rows are indexed by the pixel patch (not by the remapped block id)
      float* dst = d.stats_partial + ((lb / a.n_tiles) * a.Ncols + n0 + tid) * 2;
      dst[0] = t1;
      dst[1] = t2;
    }
  }
}

bool fast_args(const unetpp_gemm_desc* d, FastArgs& a) {
  if (d == nullptr || d->N <= 0 || d->H <= 0 || d->W <= 0) return false;
  if (d->taps != 9 && d->taps != 1) return false;
  if (d->n_in < 1 || d->n_in > UNETPP_MAX_VIEWS || d->n_out < 1 || d->n_out > UNETPP_MAX_VIEWS) return false;
  a.d = *d;
  a.Ktot = a.Ncols = a.n_tiles = a.n_chunks = 0;
  for (int i = 0; i < d->n_in; ++i) {
    const unetpp_view& v = d->in[i];
    if (!view_ok(v) || !view_covers(v, d->H, d->W)) return false;
    if (v.scale != nullptr || v.gate != nullptr || v.relu) return false;
    if (((v.C | v.c_off | v.c_len) & 3) != 0 || (reinterpret_cast<uintptr_t>(v.ptr) & 15) != 0) return false;
    if (static_cast<long>(d->N) * v.Hs * v.Ws * v.C >= 0x7fffffffL) return false;  // 32-bit element offsets
    a.Ktot += v.c_len;
    a.n_chunks += (v.c_len + KC - 1) / KC;
  }
  for (int i = 0; i < d->n_out; ++i) {
    if (!view_ok(d->out[i]) || !view_covers(d->out[i], d->H, d->W)) return false;
    a.Ncols += d->out[i].c_len;
    a.n_tiles += (d->out[i].c_len + 31) / 32;
  }
  const TileGeom g = tile_geom(d->H, d->W);
  a.log2tw = g.log2tw;
  a.tiles_x = g.tiles_x;
  a.tiles_y = g.tiles_y;
  a.total_blocks = static_cast<long>(d->N) * g.tiles_y * g.tiles_x * a.n_tiles;
  return a.total_blocks <= 0x7fffffffL;
}

}  // namespace

int launch_gemm_fast(const unetpp_gemm_desc* d, hipStream_t st) {
  FastArgs a;
  if (!fast_args(d, a) || d->weight_image == nullptr) return UNETPP_EINVAL;
  if (d->stats_partial != nullptr && d->n_out != 1) return UNETPP_EINVAL;
  const dim3 grid(static_cast<unsigned>(a.total_blocks));
  if (d->taps == 9)
    hipLaunchKernelGGL(gemm_fast_kernel<9>, grid, dim3(kThreads), 0, st, a);
  else
    hipLaunchKernelGGL(gemm_fast_kernel<1>, grid, dim3(kThreads), 0, st, a);
  return launch_status();
}

}  // namespace unetpp

using namespace unetpp;

extern "C" int64_t unetpp_gemm_weight_image_floats(const unetpp_gemm_desc* d) {
  FastArgs a;
  if (!fast_args(d, a)) return 0;
  return static_cast<int64_t>(a.n_tiles) * a.n_chunks * d->taps * 512;
}

extern "C" int unetpp_gemm_pack_weight_image(const unetpp_gemm_desc* d, float* image, void* stream) {
  FastArgs a;
  if (!fast_args(d, a) || image == nullptr || d->weight == nullptr) return UNETPP_EINVAL;
  const long total = static_cast<long>(a.n_tiles) * a.n_chunks * d->taps * 512;
  const unsigned blocks = static_cast<unsigned>((total + 255) / 256);
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (d->taps == 9)
    hipLaunchKernelGGL(pack_image_kernel<9>, dim3(blocks), dim3(256), 0, st, a, image);
  else
    hipLaunchKernelGGL(pack_image_kernel<1>, dim3(blocks), dim3(256), 0, st, a, image);
  return launch_status();
}

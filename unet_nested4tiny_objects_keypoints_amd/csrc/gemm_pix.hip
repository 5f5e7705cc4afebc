// Multi-view pixel GEMM on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32, exact fp32).
//
//   out[p, n] = bias[n] + sum_{tap, view, c} in_view[p (+) tap, c] * W[tap][k(view, c)][n]
//
// One workgroup = 256 logical pixels (TH x TW patch of one image) x 32 output columns,
// 4 waves x (2 MFMA row tiles of 32 pixels).  K is walked view by view in chunks of 16
// channels: the chunk's input patch (with its 3x3 halo, load transform applied, zero padding
// applied after the transform) and the chunk's weights for all taps are staged in LDS, then
// every tap is a pure LDS->MFMA loop.  A dense-skip concatenation is just "several views".
//
// LDS images (fp32):
//   in_tile [halo pixel][16 + 4 pad]  -- pixel stride 20 dwords: the 16 lanes of a ds_read_b128
//                                        lane group hit 16 distinct 16-byte slots
//   w_tile  [tap][k/8][col 32][8 + 4 pad] -- lane (col j, half h) reads k = 8g+4h .. +3 as one b128
// MFMA operand maps (cdna guide section 3): A lane l = (row l&31, k l>>5), B lane l = (k l>>5, col l&31),
// D reg r of lane l = row (r&3)+8*(r>>2)+4*(l>>5), col l&31.  Rows are pixels, columns are
// output channels, so one stored register is 32 consecutive floats of one pixel (a full 128-B line).
#include "bn_fused.h"
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>

#include "common.h"
#include "gemm_units.h"

namespace unetpp {
namespace {

constexpr int KC = 16;   // channels per K chunk
constexpr int KCP = 20;  // in_tile pixel stride (floats)
constexpr int WJ = 12;   // w_tile column stride (floats)

struct GemmArgs {
  unetpp_gemm_desc d;
  int log2tw, tiles_x, tiles_y;
  int Ktot, Ncols;
};

template <int TAPS>
__global__ __launch_bounds__(kThreads, 2) void gemm_pix_kernel(const GemmArgs a) {
  constexpr int HALO = (TAPS == 9) ? 1 : 0;
  constexpr int IN_FLOATS = (TAPS == 9 ? kMaxHaloPixels : kBlockPixels) * KCP;
  constexpr int W_FLOATS = TAPS * (KC / 8) * 32 * WJ;
  __shared__ __attribute__((aligned(16))) float smem[IN_FLOATS + W_FLOATS];
  float* in_tile = smem;
  float* w_tile = smem + IN_FLOATS;

  const unetpp_gemm_desc& d = a.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;

  // ---- which pixel patch, which 32 output columns ----
  int bid = blockIdx.x;
  const int txi = bid % a.tiles_x;
  bid /= a.tiles_x;
  const int tyi = bid % a.tiles_y;
  const int n = bid / a.tiles_y;
  const int TW = 1 << a.log2tw, TH = kBlockPixels >> a.log2tw;
  const int ty0 = tyi * TH, tx0 = txi * TW;
  const int HWp = TW + 2 * HALO, HHp = TH + 2 * HALO;
  const int npix = HWp * HHp;

  int ov = 0, nt = blockIdx.y, col_base = 0;
  while (ov < d.n_out - 1) {
    const int tiles_v = (d.out[ov].c_len + 31) >> 5;
    if (nt < tiles_v) break;
    nt -= tiles_v;
    col_base += d.out[ov].c_len;
    ++ov;
  }
  const unetpp_view& O = d.out[ov];
  const int n0 = col_base + nt * 32;               // first GEMM column of this tile
  const int n_cnt = min(32, O.c_len - nt * 32);    // valid columns

  int abase[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int p = 64 * wave + 32 * mt + j;
    abase[mt] = ((p >> a.log2tw) * HWp + (p & (TW - 1))) * KCP + 4 * h;
  }
  const int wbase = j * WJ + 4 * h;

  f32x16 acc[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;

  int kbase = 0;
  for (int s = 0; s < d.n_in; ++s) {
    const unetpp_view& V = d.in[s];
    const bool vec = view_vec4(V);
    for (int c0 = 0; c0 < V.c_len; c0 += KC) {
      const int c_cnt = min(KC, V.c_len - c0);
      __syncthreads();
      // ---- stage the input patch: (halo pixel, 4-channel group) items ----
      for (int it = tid; it < npix * (KC / 4); it += kThreads) {
        const int hp = it >> 2, cc = (it & 3) * 4;
        const int hy = hp / HWp, hx = hp - hy * HWp;
        const int y = ty0 + hy - HALO, x = tx0 + hx - HALO;
        f32x4 val = {0.f, 0.f, 0.f, 0.f};
        if (cc < c_cnt && y >= 0 && y < d.H && x >= 0 && x < d.W)
          val = view_load4(V, view_pixel_offset(V, n, y, x), c0 + cc, c_cnt - cc, vec);
        *reinterpret_cast<f32x4*>(&in_tile[hp * KCP + cc]) = val;
      }
      // ---- stage the weights of this chunk for every tap: (tap, k quad, column) items ----
      for (int it = tid; it < TAPS * (KC / 4) * 32; it += kThreads) {
        const int jj = it & 31, kq = (it >> 5) & 3, tap = it >> 7;
        f32x4 wv = {0.f, 0.f, 0.f, 0.f};
        if (jj < n_cnt) {
          const float* wp = d.weight + (static_cast<long>(tap) * a.Ktot + kbase + c0 + kq * 4) * a.Ncols + n0 + jj;
#pragma unroll
          for (int i = 0; i < 4; ++i)
            if (kq * 4 + i < c_cnt) wv[i] = wp[static_cast<long>(i) * a.Ncols];
        }
        *reinterpret_cast<f32x4*>(&w_tile[((tap * 2 + (kq >> 1)) * 32 + jj) * WJ + (kq & 1) * 4]) = wv;
      }
      __syncthreads();
      // ---- LDS -> MFMA ----
      const int ngroups = (c_cnt + 7) >> 3;
#pragma unroll
      for (int tap = 0; tap < TAPS; ++tap) {
        const int toff = (TAPS == 9) ? ((tap / 3) * HWp + (tap % 3)) * KCP : 0;
#pragma unroll
        for (int g = 0; g < KC / 8; ++g) {
          if (g < ngroups) {
            const f32x4 b = *reinterpret_cast<const f32x4*>(&w_tile[(tap * 2 + g) * 32 * WJ + wbase]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              const f32x4 av = *reinterpret_cast<const f32x4*>(&in_tile[abase[mt] + toff + 8 * g]);
#pragma unroll
              for (int t = 0; t < 4; ++t)
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], b[t], acc[mt], 0, 0, 0);
            }
          }
        }
      }
    }
    kbase += V.c_len;
  }

  // ---- epilogue: bias, ReLU, gate, store / accumulate, optional BatchNorm partial sums ----
  const bool col_ok = j < n_cnt;
  const float bj = (d.bias != nullptr && col_ok) ? d.bias[n0 + j] : 0.f;
  float s1 = 0.f, s2 = 0.f;
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
        s2 = fmaf(v, v, s2);  // explicit fma: the same rounding as the fast kernel's epilogue
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
    s2 += __shfl_xor(s2, 32);
    __syncthreads();  // all waves are done with the LDS tiles
    if (h == 0) {
      smem[(wave * 32 + j) * 2 + 0] = s1;
      smem[(wave * 32 + j) * 2 + 1] = s2;
    }
    __syncthreads();
    if (tid < 32 && tid < n_cnt) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        t1 += smem[(w * 32 + tid) * 2 + 0];
        t2 += smem[(w * 32 + tid) * 2 + 1];
      }
      float* dst = d.stats_partial + (static_cast<long>(blockIdx.x) * a.Ncols + n0 + tid) * 2;
      dst[0] = t1;
      dst[1] = t2;
    }
  }
}

}  // namespace
}  // namespace unetpp

namespace unetpp {
namespace {
thread_local const char* g_last_kernel = "";

// The switches of common.h's Opt: one table for the process, filled from the environment ONCE (std::call_once, at the
// first lookup) and changed afterwards only through unetpp_debug_set().  No launch path calls getenv.
const char* const kOptNames[OPT_COUNT] = {
    "BF16_NO_DMA", "BF16_DMA_ALL", "BF16_DMA_MIN8", "BF16_DMA_FORM", "BF16_DMA_SMALL", "BF16_DMA_STATS",
    "BF16_DMA_POINTWISE", "BF16_DMA_SPLIT", "BF16_WGRAD_QUAD", "WINO_NO_LEAN", "WINO_ONE_PER_CU", "MEMSET_NODES",
    "BF16_PW_PLAIN", "PW_DIRECT", "PW_NT", "HEAD_WGS_PER_CU"};
struct OptSlot {
  std::atomic<long> value{0};
  std::atomic<bool> set{false};
};
OptSlot g_opts[OPT_COUNT];
std::atomic<int> g_reserved_cus{0};
std::once_flag g_opts_once;

void opts_from_environment() {
  std::call_once(g_opts_once, [] {
    char name[64];
    for (int i = 0; i < OPT_COUNT; ++i) {
      snprintf(name, sizeof(name), "UNETPP_%s", kOptNames[i]);
      const char* e = getenv(name);
      if (e != nullptr) {
        // a variable that is present but empty / not a number counts as 1 (the old `getenv(..) != nullptr` switches)
        char* end = nullptr;
        const long v = strtol(e, &end, 10);
        g_opts[i].value.store(end == e ? 1 : v, std::memory_order_relaxed);
        g_opts[i].set.store(true, std::memory_order_release);
      }
    }
    if (const char* e = getenv("UNETPP_RESERVED_CUS"); e != nullptr) {
      const int v = atoi(e);
      g_reserved_cus.store(v < 0 ? 0 : v, std::memory_order_relaxed);
    }
  });
}
}  // namespace

bool opt_is_set(Opt o) {
  opts_from_environment();
  return g_opts[o].set.load(std::memory_order_acquire);
}
long opt_value(Opt o, long dflt) {
  opts_from_environment();
  return g_opts[o].set.load(std::memory_order_acquire) ? g_opts[o].value.load(std::memory_order_relaxed) : dflt;
}
int reserved_cus() {
  opts_from_environment();
  return g_reserved_cus.load(std::memory_order_relaxed);
}
void note_kernel(const char* name) { g_last_kernel = name; }
}  // namespace unetpp

using namespace unetpp;

extern "C" const char* unetpp_last_kernel_name(void) { return g_last_kernel; }

extern "C" int unetpp_debug_set(const char* name, int64_t value, int32_t set) {
  if (name == nullptr) return UNETPP_EINVAL;
  opts_from_environment();   // (so that a later first lookup does not overwrite this call with the environment)
  for (int i = 0; i < OPT_COUNT; ++i)
    if (strcmp(name, kOptNames[i]) == 0) {
      g_opts[i].value.store(static_cast<long>(value), std::memory_order_relaxed);
      g_opts[i].set.store(set != 0, std::memory_order_release);
      return UNETPP_OK;
    }
  return UNETPP_EINVAL;
}

extern "C" int unetpp_debug_get(const char* name, int64_t* value) {
  if (name == nullptr) return UNETPP_EINVAL;
  opts_from_environment();
  for (int i = 0; i < OPT_COUNT; ++i)
    if (strcmp(name, kOptNames[i]) == 0) {
      if (!g_opts[i].set.load(std::memory_order_acquire)) return 0;
      if (value != nullptr) *value = g_opts[i].value.load(std::memory_order_relaxed);
      return 1;
    }
  return UNETPP_EINVAL;
}

extern "C" int32_t unetpp_set_reserved_cus(int32_t n) {
  opts_from_environment();
  if (n >= 0) {
    const int cus = physical_cu_count();
    const int cap = cus > 8 ? cus - 8 : 0;
    g_reserved_cus.store(n > cap ? cap : n, std::memory_order_relaxed);
  }
  return reserved_cus();
}

extern "C" int32_t unetpp_usable_cus(int32_t* physical) {
  if (physical != nullptr) *physical = physical_cu_count();
  return device_cu_count();
}

extern "C" int64_t unetpp_gemm_pixel_blocks(int32_t N, int32_t H, int32_t W) {
  if (N <= 0 || H <= 0 || W <= 0) return 0;
  const TileGeom g = tile_geom(H, W);
  return static_cast<int64_t>(N) * g.tiles_y * g.tiles_x;
}

extern "C" int64_t unetpp_gemm_stats_rows(int32_t N, int32_t H, int32_t W) {
  const int64_t blocks = unetpp_gemm_pixel_blocks(N, H, W);
  if (blocks <= 0) return 0;
  return blocks > kBnFusedRows ? blocks : kBnFusedRows;  // per-workgroup rows (bn_fused.h) or per-block rows
}

namespace {
// *bn_rows: rows of BatchNorm partial sums the launched kernel writes when that is NOT one per 256-pixel block (the
// persistent kernels of bn_fused.h write one per workgroup); left untouched otherwise
int gemm_fwd_dispatch(const unetpp_gemm_desc* d, void* stream, long* bn_rows);
}

extern "C" int unetpp_gemm_fwd(const unetpp_gemm_desc* d, void* stream) {
  if (d == nullptr || d->N <= 0 || d->H <= 0 || d->W <= 0) return UNETPP_EINVAL;
  const unetpp_bn_fused& bn = d->bn;
  long bn_rows = 0;
  if (bn.scale == nullptr) return gemm_fwd_dispatch(d, stream, &bn_rows);
  // BatchNorm finalize attached to this call: over the kernel's per-workgroup rows where it writes those (bn_fused.h),
  // else over per-block rows
  if (!d->stats_partial || !bn.gamma || !bn.beta || !bn.mean || !bn.invstd || !bn.shift || bn.count < 1 ||
      (bn.running_mean == nullptr) != (bn.running_var == nullptr) || d->n_out != 1)
    return UNETPP_EINVAL;
  const int rc = gemm_fwd_dispatch(d, stream, &bn_rows);
  if (rc != UNETPP_OK) return rc;
  const int64_t rows = bn_rows > 0 ? bn_rows : unetpp_gemm_pixel_blocks(d->N, d->H, d->W);
  return unetpp_bn_finalize(d->stats_partial, rows, d->out[0].c_len, bn.count, bn.gamma, bn.beta, bn.eps, bn.momentum,
                            bn.running_mean, bn.running_var, bn.mean, bn.invstd, bn.scale, bn.shift, stream);
}

namespace {
int gemm_fwd_dispatch(const unetpp_gemm_desc* d, void* stream, long* bn_rows) {
  if (d == nullptr || d->N <= 0 || d->H <= 0 || d->W <= 0) return UNETPP_EINVAL;
  if (d->taps != 9 && d->taps != 1) return UNETPP_EINVAL;
  if (d->n_in < 1 || d->n_in > UNETPP_MAX_VIEWS || d->n_out < 1 || d->n_out > UNETPP_MAX_VIEWS) return UNETPP_EINVAL;
  if (d->weight == nullptr && d->weight_image == nullptr) return UNETPP_EINVAL;
  if (d->stats_partial != nullptr && d->n_out != 1) return UNETPP_EINVAL;
  if (d->flags & UNETPP_GEMM_BF16) {  // bf16 storage: the MFMA kernel, or the VALU first layer (fp32 input, bf16 output)
    if (d->weight_image != nullptr) {
      if (d->taps == 1) {  // plain pointwise launches whose weights fit LDS: gemm_pw_bf16.hip
        FastArgs fa;
        if (bf16_gemm_args(d, fa)) {
          const int pw = launch_gemm_pw_bf16(d, fa, static_cast<hipStream_t>(stream));
          if (pw != 1) return pw;
        }
      }
      const int dma = launch_gemm_bf16_dma(d, static_cast<hipStream_t>(stream));
      return dma != 1 ? dma : launch_gemm_bf16(d, static_cast<hipStream_t>(stream));
    }
    const int small = launch_small_cin_fwd(d, static_cast<hipStream_t>(stream), bn_rows);
    return small == 1 ? UNETPP_EINVAL : small;  // no generic bf16 kernel: unaligned views are refused
  }
  if (d->weight_image != nullptr)  // the image was packed for the algorithm the same descriptor selects
    return wino_applies(d) ? launch_gemm_wino(d, static_cast<hipStream_t>(stream), bn_rows)
                           : launch_gemm_fast(d, static_cast<hipStream_t>(stream));
  GemmArgs a;
  a.d = *d;
  a.Ktot = 0;
  a.Ncols = 0;
  int n_tiles = 0;
  for (int i = 0; i < d->n_in; ++i) {
    if (!view_ok(d->in[i]) || !view_covers(d->in[i], d->H, d->W)) return UNETPP_EINVAL;
    a.Ktot += d->in[i].c_len;
  }
  for (int i = 0; i < d->n_out; ++i) {
    if (!view_ok(d->out[i]) || !view_covers(d->out[i], d->H, d->W)) return UNETPP_EINVAL;
    a.Ncols += d->out[i].c_len;
    n_tiles += (d->out[i].c_len + 31) / 32;
  }
  const TileGeom g = tile_geom(d->H, d->W);
  a.log2tw = g.log2tw;
  a.tiles_x = g.tiles_x;
  a.tiles_y = g.tiles_y;
  const int64_t pix_blocks = static_cast<int64_t>(d->N) * g.tiles_y * g.tiles_x;
  if (pix_blocks > 0x7fffffffLL || n_tiles > 65535) return UNETPP_EINVAL;
  {
    const int small = launch_small_cin_fwd(d, static_cast<hipStream_t>(stream), bn_rows);  // 1..4-channel first layer
    if (small != 1) return small;
  }
  const dim3 grid(static_cast<unsigned>(pix_blocks), static_cast<unsigned>(n_tiles));
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (d->taps == 9)
    hipLaunchKernelGGL(gemm_pix_kernel<9>, grid, dim3(kThreads), 0, st, a);
  else
    hipLaunchKernelGGL(gemm_pix_kernel<1>, grid, dim3(kThreads), 0, st, a);
  note_kernel(d->taps == 9 ? "gemm_pix_kernel<9>" : "gemm_pix_kernel<1>");
  return launch_status();
}
}  // namespace

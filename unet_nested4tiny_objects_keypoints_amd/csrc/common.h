// Shared device/host helpers for the gfx950 UNet_Nested kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "unetpp_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace unetpp {

// Pixel tile of the MFMA kernels: 256 logical pixels per workgroup, laid out TH x TW with
// TW a power of two in {8, 16, 32}; 4 waves, each owning 64 of those pixels.
constexpr int kBlockPixels = 256;
constexpr int kThreads = 256;
constexpr int kMaxHaloPixels = 340;  // max over TW in {8,16,32} of (TH+2)*(TW+2)

struct TileGeom {
  int log2tw, tiles_x, tiles_y;
};

inline TileGeom tile_geom(int H, int W) {
  int l = 3;
  while ((1 << l) < W && l < 5) ++l;
  TileGeom g;
  g.log2tw = l;
  const int TW = 1 << l, TH = kBlockPixels >> l;
  g.tiles_x = (W + TW - 1) / TW;
  g.tiles_y = (H + TH - 1) / TH;
  return g;
}

// Value of lane (lane ^ OFF), OFF a compile-time power of two below 32, without the LDS crossbar address path of
// __shfl_xor (v_mbcnt + shift + ds_bpermute_b32 + wait): OFF 1 and 2 are DPP quad permutations (a VALU move), 4..16 a
// ds_swizzle in bit mode (no address register).
template <int OFF>
__device__ __forceinline__ float xor_lane(float v) {
  static_assert(OFF == 1 || OFF == 2 || OFF == 4 || OFF == 8 || OFF == 16, "xor_lane: 1, 2, 4, 8 or 16");
  const int i = __builtin_bit_cast(int, v);
  int r;
  if constexpr (OFF == 1) r = __builtin_amdgcn_mov_dpp(i, 0xB1, 0xF, 0xF, true);        // quad_perm [1, 0, 3, 2]
  else if constexpr (OFF == 2) r = __builtin_amdgcn_mov_dpp(i, 0x4E, 0xF, 0xF, true);   // quad_perm [2, 3, 0, 1]
  else r = __builtin_amdgcn_ds_swizzle(i, 0x1F | (OFF << 10));                            // bit mode: and 0x1F, xor OFF
  return __builtin_bit_cast(float, r);
}

// Sum over the 16 lanes of a DPP row (lanes 16 r .. 16 r + 15) in a fixed order; every lane of the row ends with the sum.
// Four VALU instructions, no LDS crossbar: quad permutations, then the half-row and the row mirrored.
__device__ __forceinline__ float row16_sum(float v) {
  v += xor_lane<1>(v);
  v += xor_lane<2>(v);
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));  // row_half_mirror
  v += __builtin_bit_cast(float, __builtin_amdgcn_mov_dpp(__builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));  // row_mirror
  return v;
}

// slab groups of the weight-gradient finish kernels for n_split slabs: the smallest power of two >= n_split, at most 16
inline int finish_log2_groups(int n_split) {
  int l = 0;
  while ((1 << l) < n_split && l < 4) ++l;
  return l;
}

inline bool view_ok(const unetpp_view& v, bool need_ptr = true) {
  if (need_ptr && v.ptr == nullptr) return false;
  if (v.C <= 0 || v.c_len <= 0 || v.c_off < 0 || v.c_off + v.c_len > v.C) return false;
  if (v.Hs <= 0 || v.Ws <= 0 || v.sy <= 0 || v.sx <= 0 || v.oy < 0 || v.ox < 0) return false;
  if ((v.scale == nullptr) != (v.shift == nullptr)) return false;
  return true;
}

// every logical pixel of an H x W grid must land inside the view's tensor
inline bool view_covers(const unetpp_view& v, int H, int W) {
  return (H - 1) * v.sy + v.oy < v.Hs && (W - 1) * v.sx + v.ox < v.Ws;
}

__device__ __forceinline__ bool view_vec4(const unetpp_view& v) {
  return ((v.C | v.c_off) & 3) == 0 && ((reinterpret_cast<uintptr_t>(v.ptr) & 15) == 0) &&
         (v.gate == nullptr || (reinterpret_cast<uintptr_t>(v.gate) & 15) == 0);
}

__device__ __forceinline__ long view_pixel_offset(const unetpp_view& v, int n, int y, int x) {
  return ((static_cast<long>(n) * v.Hs + (y * v.sy + v.oy)) * v.Ws + (x * v.sx + v.ox)) * v.C + v.c_off;
}

// Load 4 consecutive slice channels [c, c+4) of one pixel with the view's load transform.
// `rem` = channels still valid from c (entries >= rem come back as 0).
__device__ __forceinline__ f32x4 view_load4(const unetpp_view& v, long off, int c, int rem, bool vec) {
  f32x4 val = {0.f, 0.f, 0.f, 0.f};
  const float* p = v.ptr + off + c;
  if (vec && rem >= 4) {
    val = *reinterpret_cast<const f32x4*>(p);
  } else {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < rem) val[i] = p[i];
  }
  if (v.scale != nullptr) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < rem) val[i] = fmaf(val[i], v.scale[c + i], v.shift[c + i]);
  }
  if (v.relu) {
#pragma unroll
    for (int i = 0; i < 4; ++i) val[i] = fmaxf(val[i], 0.f);
  }
  if (v.gate != nullptr) {
    const float* g = v.gate + off + c;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      if (i < rem) val[i] = (g[i] > 0.f) ? val[i] : 0.f;
  }
  return val;
}

inline int launch_status() { return hipGetLastError() == hipSuccess ? UNETPP_OK : UNETPP_ELAUNCH; }

// CUs of the current device, queried once per device (every persistent-grid launcher sizes its grid from it: ~100
// launches per step); 0 when the runtime cannot tell
int reserved_cus();  // unetpp_set_reserved_cus (gemm_pix.hip): CUs the persistent grids leave to a concurrent collective
inline int physical_cu_count() {
  static int cached[64] = {};   // (benign race: every writer stores the same value)
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0) return 0;
  if (dev < 64 && cached[dev] > 0) return cached[dev];
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) return 0;
  if (dev < 64) cached[dev] = cus;
  return cus;
}
inline int device_cu_count() {  // what a persistent grid may fill
  const int cus = physical_cu_count();
  if (cus <= 0) return 0;
  const int left = cus - reserved_cus();
  return left < 8 ? (cus < 8 ? cus : 8) : left;
}

// remembers the kernel a dispatch chose (unetpp_last_kernel_name); defined in gemm_pix.hip
void note_kernel(const char* name);

// A/B and test switches of the dispatchers (unetpp_debug_set, gemm_pix.hip).  A switch is an int that is either unset
// (the dispatcher's built-in default applies) or set -- by unetpp_debug_set(), else by the environment variable
// UNETPP_<name>, which is read ONCE per process when the first switch is looked up (never per launch).
enum Opt {
  OPT_BF16_NO_DMA,         // 1: the bf16 GEMMs never take the LDS-DMA kernel (tests compare the two kernels)
  OPT_BF16_DMA_ALL,        // 1: every launch the LDS-DMA kernel can express takes it
  OPT_BF16_DMA_MIN8,       // 8-wave form when at least this many 512-pixel units per CU exist (default 2)
  OPT_BF16_DMA_FORM,       // 0: no DMA kernel, 4 / 8: force that form where it applies
  OPT_BF16_DMA_SMALL,      // 0: small 3x3 launches back on the register kernel (round-3 dispatch)
  OPT_BF16_DMA_STATS,      // 0: BatchNorm-statistics launches back on the register kernel
  OPT_BF16_DMA_POINTWISE,  // 0: transposed-convolution GEMMs back on the register kernel
  OPT_BF16_DMA_SPLIT,      // 1: split-tail second launch (experiment, off)
  OPT_BF16_WGRAD_QUAD,     // 0: wide bf16 weight gradients on the pair kernel
  OPT_WINO_NO_LEAN,        // 1: the general Winograd instantiation for every launch
  OPT_WINO_ONE_PER_CU,     // 1: (stamped / experiment builds) one Winograd workgroup per CU
  OPT_MEMSET_NODES,        // 1: unused workspace rows cleared by hipMemsetAsync instead of zero_rows_kernel (graph probe only)
  OPT_BF16_PW_PLAIN,       // 0: plain-output bf16 pointwise launches on the two-per-CU instantiation (round-4 form)
  OPT_PW_DIRECT,           // 0: fp32 pointwise GEMMs back on gemm_fast_kernel<1> (tests compare the two kernels)
  OPT_PW_NT,               // fp32 pointwise GEMM: 1 / 0 = non-temporal / plain stores whatever the output size
  OPT_HEAD_WGS_PER_CU,     // bf16 head backward: at most this many workgroups per CU take tiles (default 4; 0 = every workgroup of the grid, as until round 6)
  OPT_COUNT
};
bool opt_is_set(Opt o);
long opt_value(Opt o, long dflt);   // dflt when unset

// gemm_fast.hip: register-prefetched kernel for plain aligned views (needs d->weight_image)
int launch_gemm_fast(const unetpp_gemm_desc* d, hipStream_t st);
// gemm_pw.hip: fp32 pointwise GEMM with the weights resident in LDS and the activations loaded straight into the MFMA
// operand registers (fa = the descriptor's fast_args); returns 1 when the descriptor is not one it takes
int launch_gemm_pw(const unetpp_gemm_desc* d, const struct FastArgs& fa, hipStream_t st);
// gemm_pw_bf16.hip: the bf16-storage twin of gemm_pw.hip (fa = bf16_gemm_args of the descriptor)
int launch_gemm_pw_bf16(const unetpp_gemm_desc* d, const struct FastArgs& fa, hipStream_t st);
// gemm_wino.hip: Winograd F(2x2,3x3) kernel for taps == 9 without UNETPP_GEMM_DIRECT (needs its own weight image)
bool wino_applies(const unetpp_gemm_desc* d);
int launch_gemm_wino(const unetpp_gemm_desc* d, hipStream_t st, long* bn_rows);  // *bn_rows = rows of BatchNorm sums written, when per workgroup
// gemm_bf16.hip: bf16-storage direct implicit GEMM (UNETPP_GEMM_BF16); needs its own weight image
bool bf16_gemm_args(const unetpp_gemm_desc* d, struct FastArgs& a);
int launch_gemm_bf16(const unetpp_gemm_desc* d, hipStream_t st);
// gemm_bf16_dma.hip: the same GEMM with both operands staged by LDS-DMA (plain input views, 32-channel slices); returns 1
// when the descriptor is not one it takes (same weight image as gemm_bf16.hip)
int launch_gemm_bf16_dma(const unetpp_gemm_desc* d, hipStream_t st);
// wgrad_bf16.hip: bf16-storage weight gradient (UNETPP_GEMM_BF16); UNETPP_EINVAL when the views do not fit
int launch_wgrad_bf16(const unetpp_wgrad_desc* d, int Ktot, int Ncols, int n_tiles_cols, int k_tiles, hipStream_t st);
bool wgrad_bf16_quads(const unetpp_wgrad_desc* d);  // the bf16 kernel will give a workgroup 2 x 2 (channel, column) tile pairs
// wgrad_fast.hip: 8-wave double-buffered kernel for plain aligned views; returns 1 when it does not apply
int launch_wgrad_fast(const unetpp_wgrad_desc* d, int Ktot, int Ncols, int n_tiles_cols, int k_tiles, hipStream_t st);
// wgrad_dma.hip: LDS-DMA staged kernel for views without load transforms; returns 1 when it does not apply
int launch_wgrad_dma(const unetpp_wgrad_desc* d, int Ktot, int Ncols, int n_tiles_cols, int k_tiles, hipStream_t st);
// wgrad_pw.hip: pointwise fp32 weight gradient with both operands loaded straight into the MFMA operand registers (64-channel
// x 128-column blocks of dW per wave); launch returns 1 when it does not apply, wgrad_pw_pairs 0
int wgrad_pw_pairs(const unetpp_wgrad_desc* d);
int launch_wgrad_pw(const unetpp_wgrad_desc* d, hipStream_t st);
// wgrad_wino.hip: Winograd F(2x2,3x3) weight gradient (16 transform-domain planes per slab); launch returns 1 when
// it does not apply
bool wgrad_wino_applies(const unetpp_wgrad_desc* d);
int launch_wgrad_wino(const unetpp_wgrad_desc* d, int Ktot, int Ncols, int n_tiles_cols, int k_tiles, hipStream_t st);
int launch_wgrad_finish_wino(const float* slabs, int n_split, int K, int Ncols, float* dw, long d_t, long d_k, long d_n,
                             float* db, hipStream_t st);
// first_layer.hip: VALU kernels for the 1..4-channel first convolution; return 1 when they do not apply
int launch_small_cin_fwd(const unetpp_gemm_desc* d, hipStream_t st, long* bn_rows);
int launch_small_cin_wgrad(const unetpp_wgrad_desc* d, hipStream_t st);

}  // namespace unetpp

// Winograd F(2x2, 3x3) form of the 3x3 multi-view pixel GEMM (forward and input gradient of every 3x3 convolution
// with 4-aligned channel slices): 16 multiplies per 2x2 output tile and (channel, column) pair instead of 36, i.e.
// 2.25x fewer MFMA cycles than gemm_fast.hip for the same algorithmic FLOPs.  fp32 throughout; against a direct
// fp32 convolution the result differs by rounding only (max error ~2x that of the direct sum).
//
//   Y = A^T [ sum_c (G g G^T) (.) (B^T d B) ] A        d = 4x4 input window, g = 3x3 filter, Y = 2x2 outputs
//
// Mapping onto v_mfma_f32_16x16x4_f32 (exact fp32, same FLOP rate as 32x32x2):
//   * unit of work = 256-pixel patch (64 tiles of 2x2) x 32 columns, 4 waves; a wave owns 16 tiles;
//   * MFMA row = tile, MFMA k = 4 channels, MFMA column = output column: per transform position xi one accumulator
//     pair (2 x 16 columns), acc[16][2] float4 = 128 VGPRs per lane.  (16 columns per unit halve that, but the input
//     transform and the epilogue then cost twice the VALU per MFMA, and VALU issue is what bounds this kernel.)
//   * A operand: lane (tile t, k-slot g) reads its 4x4 window one channel at a time (16 ds_read_b32), runs the input
//     transform B^T d B IN REGISTERS (32 add/sub per channel) and feeds the 16 results straight to 32 MFMAs -- the
//     transformed input never exists in memory;
//   * B operand: the transformed weights U = G g G^T are precomputed per launch (weight_image.hip) as an LDS image
//     whose lane-linear 8-byte reads each feed two MFMAs;
//   * all 16 xi of a (tile, column) land in the same lane and register index, so the output transform A^T M A is
//     lane-local (24 add/sub per 2x2 tile); bias/ReLU/gate/accumulate/BatchNorm partial sums and 16-byte transposed
//     stores follow as in the direct kernel;
//   * K in chunks of 8 channels in a persistent, flattened (unit, chunk) stream: the inputs go global -> registers
//     (two chunks ahead; BatchNorm fold, ReLU and zero padding are applied on the way) -> LDS, the weight image goes
//     L2 -> LDS by LDS-DMA one chunk ahead; double-buffered LDS, one barrier per chunk; a workgroup walks a CONTIGUOUS
//     run of units (column tile fastest) with a scalar cursor, so the patch geometry (offsets, bounds mask) is only
//     recomputed when the pixel patch changes; 65 KB LDS, two workgroups per CU;
//   * the LDS operand reads of the MFMA phase are placed by hand (inline asm, one group of 8 MFMAs ahead, one
//     s_waitcnt per group, scheduling fences around each group): hipcc sinks every ds_read to just in front of its
//     first use, which leaves a lone wave's matrix pipe idle for an LDS round trip 13 times per 32 MFMAs.
//   * LEAN instantiations (MODE 1, 2: every input view an fp32 tensor of the launch geometry, channel slices in
//     multiples of 8, at most 2 GB): the inputs come in through buffer loads whose hardware range check returns zeros
//     for out-of-image halo pixels (their offset is simply out of range), so the staging of a chunk is three loads and
//     three LDS stores per thread -- no clamping, no selects -- issued inside the first MFMA groups of the chunk
//     before.  MODE 2 adds the BatchNorm fold / ReLU on load: the chunk's coefficients are requested with the cursor,
//     a chunk ahead, and the padding is restored by a bitwise AND with the in-image mask.  MODE 0 is the general
//     kernel (any view geometry, partial chunks, per-element padding select).
#include <cstdlib>
#include <utility>

#include "bn_fused.h"
#include "common.h"
#include "gemm_units.h"
#include "lds_asm.h"
#include "wino_experiments.h"

namespace unetpp {
namespace {

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4v __attribute__((vector_size(16)));  // the buffer-load builtin's own return type
constexpr unsigned kOutOfRange = 0x80000000u;  // buffer offset no LEAN view reaches (views are at most 2 GB)
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// input transform B^T d B of one 4x4 window (row major), 32 add/sub
__device__ __forceinline__ void wino_input_transform(const float (&dd)[16], float (&V)[16]) {
  float t[4][4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    t[0][j] = dd[0 + j] - dd[8 + j];
    t[1][j] = dd[4 + j] + dd[8 + j];
    t[2][j] = dd[8 + j] - dd[4 + j];
    t[3][j] = dd[4 + j] - dd[12 + j];
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    V[4 * i + 0] = t[i][0] - t[i][2];
    V[4 * i + 1] = t[i][1] + t[i][2];
    V[4 * i + 2] = t[i][2] - t[i][1];
    V[4 * i + 3] = t[i][1] - t[i][3];
  }
}

// the same on register pairs: dd[2 * i + p] = (d[i][2p], d[i][2p + 1]); 16 packed adds, bit-identical results
__device__ __forceinline__ void wino_input_transform_pk(const f32x2_t (&dd)[8], float (&V)[16]) {
  f32x2_t t[8];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    t[0 + p] = pk_sub(dd[0 + p], dd[4 + p]);  // row 0 - row 2
    t[2 + p] = pk_add(dd[2 + p], dd[4 + p]);  // row 1 + row 2
    t[4 + p] = pk_sub(dd[4 + p], dd[2 + p]);  // row 2 - row 1
    t[6 + p] = pk_sub(dd[2 + p], dd[6 + p]);  // row 1 - row 3
  }
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f32x2_t va = pk_sub_add_x(t[2 * i], t[2 * i + 1]);  // (t0 - t2, t1 + t2)
    const f32x2_t vb = pk_cross_sub(t[2 * i], t[2 * i + 1]);  // (t2 - t1, t1 - t3)
    V[4 * i + 0] = va[0];
    V[4 * i + 1] = va[1];
    V[4 * i + 2] = vb[0];
    V[4 * i + 3] = vb[1];
  }
}

constexpr int WKC = 8;      // channels per K chunk
constexpr int WNC = 32;     // columns per unit
constexpr int WP = 10;      // LDS pixel stride of the input patch (floats): 8 channels + 2 pad
constexpr int WIMG = 4096;  // floats of one (column tile, chunk) weight image

// In-kernel phase stamps (profiling builds only: -DUNETPP_WINO_STAMPS, see tools/wino_stamps.py): every wave adds the
// s_memtime cycles it spent in each phase of its (unit, chunk) stream to a global table.
#define WINO_FENCE() __builtin_amdgcn_sched_barrier(0)
#ifdef UNETPP_WINO_STAMPS
__device__ unsigned long long g_wino_stamps[16];
#define WINO_STAMP(i)                          \
  do {                                         \
    const unsigned long long now_ = clock64(); \
    st_acc[i] += now_ - st_last;               \
    st_last = now_;                            \
  } while (0)
#else
#define WINO_STAMP(i) \
  do {                \
  } while (0)
#endif

// NH = 16-column halves of the tile that are computed: 2, or 1 when no output view is wider than 16 channels (narrow
// networks, e.g. the reference's default base width 16) -- the second half would multiply zero weights.
// BNF: the BatchNorm partial sums go out as ONE row per workgroup (bn_fused.h); an instantiation of its own, so that the
// 30 launches of a step without statistics keep their registers (a runtime switch cost 8 more scalar spills and 5 %)
template <int LOG2TW, int NH, int MODE, bool BNF = false>
__global__ __launch_bounds__(kThreads, 2) void gemm_wino_kernel(const FastArgs a) {
  constexpr bool LEAN = MODE != 0, FOLD = MODE == 2;
  constexpr int TW = 1 << LOG2TW, TH = kBlockPixels >> LOG2TW;
  constexpr int HWp = TW + 2, HHp = TH + 2;
  constexpr int NPIX = HWp * HHp;
  constexpr int TXN = TW / 2;        // 2x2 tiles per patch row (a power of two)
  constexpr int IN_FLOATS = 4096;    // >= kMaxHaloPixels * WP (3400) and the dummy staging items behind the patch
  constexpr int IN_ITEMS = (NPIX * 2 + kThreads - 1) / kThreads;  // 16-byte items, 2 per pixel (<= 3)
  constexpr int W_ITEMS = WIMG / 4 / kThreads;                    // 4
  static_assert(IN_ITEMS == 3 && W_ITEMS == 4, "the staging lambdas' default ranges");
  static_assert(kMaxHaloPixels * WP <= IN_FLOATS, "input patch does not fit");
  static_assert(((IN_ITEMS * kThreads - 1) >> 1) * WP + 8 <= IN_FLOATS, "staging items past the patch must stay inside the buffer");
  // Two (input patch, weight image) buffers: the next chunk is written into the other buffer at the top of the
  // current chunk (its loads were issued a whole chunk earlier), so a chunk costs one barrier.
  constexpr int BUF = IN_FLOATS + WIMG;
  __shared__ __attribute__((aligned(16))) float smem[2 * BUF];
  float* in_tile = smem;             // buffers of the chunk being computed
  float* w_tile = smem + IN_FLOATS;
  // the four waves' BatchNorm partial sums of the unit just finished; then (BNF) the sums of ALL units of this workgroup
  // per column
  constexpr int RUN0 = 4 * WNC * 2;
  __shared__ float stat_lds[BNF ? RUN0 + kBnFusedMaxCols * 2 : RUN0];
  // The launch's bias vector (zeros past Ncols / without a bias), read by every epilogue: from global memory that read
  // sat at the head of the epilogue with a full memory round trip in front of the first output value.
  constexpr int kBiasCols = 512;
  __shared__ __attribute__((aligned(16))) float bias_lds[kBiasCols];

  const unetpp_gemm_desc& d = a.d;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, t16 = lane & 15, g = lane >> 4;

  const UnitRange ur = my_contiguous_unit_range(a.total_blocks);
  for (int i = tid; i < kBiasCols; i += kThreads)  // (a barrier follows in the prologue)
    bias_lds[i] = (a.d.bias != nullptr && i < a.Ncols) ? a.d.bias[i] : 0.f;
  constexpr bool bn_fused = BNF;
  if constexpr (BNF) {
    for (int i = tid; i < kBnFusedMaxCols * 2; i += kThreads) stat_lds[RUN0 + i] = 0.f;  // (a barrier follows in the prologue)
  }
  if (ur.count == 0) {  // (cannot happen with the launcher's grids; the workgroup's row must exist all the same)
    if constexpr (BNF) {
      __syncthreads();
      bn_rows_store<kThreads>(a.d.stats_partial, a.Ncols, stat_lds + RUN0);
    }
    return;
  }
  // Everything that steers the (unit, chunk) stream is wave uniform; readfirstlane moves it to scalar registers
  // (hipcc computes the range with vector divisions and would keep the whole cursor in VGPRs otherwise).
  const int n_units = __builtin_amdgcn_readfirstlane(static_cast<int>(ur.count));

  // compute side: this lane's tile (A operand / input transform) and weight-image slot (B operand)
  const int my_tile = 16 * wave + t16;
  const int a_base = ((2 * (my_tile / TXN)) * HWp + 2 * (my_tile % TXN)) * WP + 2 * g;  // channel 2g (+s)
  const int b_base = (g * 16 + t16) * 2;

  f32x4 acc[16][NH];
#pragma unroll
  for (int xi = 0; xi < 16; ++xi)
#pragma unroll
    for (int nh = 0; nh < NH; ++nh) acc[xi][nh] = f32x4{0.f, 0.f, 0.f, 0.f};

  // ---- prefetch side: the cursor describes the chunk whose inputs sit in reg_in (two chunks ahead of the MFMAs) ----
  f32x4 reg_in[IN_ITEMS];
  unsigned voff[IN_ITEMS];
  unsigned in_mask = 0;
  int pf_cnt = 0;
  int p_unit = 0;
  int p_s = 0, p_c0 = 0, p_chunk = 0;
  int p_n = 0, p_ty0 = 0, p_tx0 = 0;
  const float* p_wimg = nullptr;
  const int cc = (tid & 1) << 2;  // channel quad of every staging item of this thread (kThreads is even)
  // the cursor's input view, cached (d.in[p_s] is a kernarg read with a scalar-cache round trip at every use)
  const float* v_ptr = nullptr;
  const float* v_scale = nullptr;
  const float* v_shift = nullptr;
  int v_clen = 0;
  bool v_relu = false;
  // LEAN: the view as a buffer resource (range-checked loads), its pixel pitch and slice origin in bytes, and the
  // linear index of every staging item's pixel in the (plain, launch-sized) input tensors, -1 outside the image
  int v_bytes = 0, v_pitch = 0, v_origin = 0;
  int pix[IN_ITEMS];
  auto cache_view = [&]() {
    const unetpp_view& V = d.in[p_s];
    v_clen = V.c_len;
    if constexpr (LEAN) {
      v_ptr = V.ptr;
      v_bytes = d.N * V.Hs * V.Ws * V.C * 4;  // <= 2 GB (launcher)
      v_pitch = V.C * 4;
      v_origin = V.c_off * 4;
      if constexpr (FOLD) {
        v_scale = V.scale;
        v_shift = V.shift;
        v_relu = V.relu != 0;
      }
    } else {
      v_ptr = V.ptr;
      v_scale = V.scale;
      v_shift = V.shift;
      v_relu = V.relu != 0;
    }
  };

  // The unit run is contiguous, so both cursors (prefetch side, compute side) are decoded once and then stepped:
  // column group fastest, then the patch along x, y, image -- scalar adds and compares instead of four divisions.
  auto step_unit = [&](UnitGeom& u) {
    if (++u.group < a.n_groups) return;
    u.group = 0;
    ++u.patch;
    u.tx0 += TW;
    if (u.tx0 < a.tiles_x * TW) return;
    u.tx0 = 0;
    u.ty0 += TH;
    if (u.ty0 < a.tiles_y * TH) return;
    u.ty0 = 0;
    ++u.n;
  };
  UnitGeom p_ug = decode_unit<LOG2TW>(a, ur.first);
  p_ug.n = __builtin_amdgcn_readfirstlane(p_ug.n);
  p_ug.ty0 = __builtin_amdgcn_readfirstlane(p_ug.ty0);
  p_ug.tx0 = __builtin_amdgcn_readfirstlane(p_ug.tx0);
  p_ug.group = __builtin_amdgcn_readfirstlane(p_ug.group);
  p_ug.patch = __builtin_amdgcn_readfirstlane(static_cast<int>(p_ug.patch));  // < 2^31 (fast_args)
  UnitGeom c_ug = p_ug;

  long p_patch = -1;
  auto prefetch_unit = [&]() -> bool {  // returns true when the pixel patch is the previous unit's
    const UnitGeom& ug = p_ug;
    p_wimg = d.weight_image + static_cast<long>(ug.group) * a.n_chunks * WIMG;
    if (ug.patch == p_patch) return true;
    p_patch = ug.patch;
    p_n = ug.n;
    p_ty0 = ug.ty0;
    p_tx0 = ug.tx0;
    if constexpr (LEAN) {
      const int row0 = p_n * d.H;
#pragma unroll
      for (int q = 0; q < IN_ITEMS; ++q) {
        const int it = tid + q * kThreads;
        const int hp = it >> 1;
        const int hy = hp / HWp, hx = hp - hy * HWp;
        const int y = p_ty0 + hy - 1, x = p_tx0 + hx - 1;
        // bitwise, and the index computed on both sides: a short-circuit here becomes a divergent branch that the
        // optimiser threads into the cursor code below, after which the whole cursor lives in vector registers
        // (only the last pass has items past the patch: for the others the test is a compile-time `true`, not a lane mask
        // that lives in -- and is spilled from -- a scalar register pair across the unit loop)
        const bool in_patch = ((q + 1) * kThreads <= NPIX * 2) ? true : (it < NPIX * 2);
        const bool inside = in_patch & (static_cast<unsigned>(y) < static_cast<unsigned>(d.H)) &
                            (static_cast<unsigned>(x) < static_cast<unsigned>(d.W));
        const int linear = (row0 + y) * d.W + x;
        pix[q] = inside ? linear : -1;
      }
      return false;
    }
    in_mask = 0;
#pragma unroll
    for (int q = 0; q < IN_ITEMS; ++q) {
      const int it = tid + q * kThreads;
      const int hp = it >> 1;
      const int hy = hp / HWp, hx = hp - hy * HWp;
      const int y = p_ty0 + hy - 1, x = p_tx0 + hx - 1;
      if ((it < NPIX * 2) && y >= 0 && y < d.H && x >= 0 && x < d.W) in_mask |= 1u << q;
    }
    return false;
  };
  auto view_offsets = [&](const unetpp_view& V) {  // clamped: every item loads from a valid address
    if constexpr (LEAN) {  // byte offsets; padding pixels and the dummy items of the last pass are out of range = zeros
#pragma unroll
      for (int q = 0; q < IN_ITEMS; ++q)
        voff[q] = static_cast<unsigned>(pix[q] * v_pitch + v_origin + cc * 4) |
                  (static_cast<unsigned>(pix[q] >> 31) & kOutOfRange);  // arithmetic on purpose: see prefetch_unit
      return;
    }
#pragma unroll
    for (int q = 0; q < IN_ITEMS; ++q) {
      const int hp = min((tid + q * kThreads) >> 1, NPIX - 1);
      const int hy = hp / HWp, hx = hp - hy * HWp;
      const int yy = min(max(p_ty0 + hy - 1, 0), d.H - 1), xx = min(max(p_tx0 + hx - 1, 0), d.W - 1);
      voff[q] = view_pixel_offset32(V, p_n, yy, xx);
    }
  };
  // next chunk of the unit, or chunk 0 of the next unit; stays on the last chunk when nothing is left (the loads and
  // stores below are unconditional, straight-line code: a load under a branch makes hipcc drain vmcnt at the join)
  auto advance = [&]() {
    if (p_chunk + 1 < a.n_chunks) {
      ++p_chunk;
      p_c0 += WKC;
      if (p_c0 >= v_clen) {
        ++p_s;
        p_c0 = 0;
        cache_view();
        view_offsets(d.in[p_s]);
      }
    } else if (p_unit + 1 < n_units) {
      ++p_unit;
      p_chunk = 0;
      p_c0 = 0;
      const bool same_view = p_s == 0;
      if (!same_view) {
        p_s = 0;
        cache_view();
      }
      step_unit(p_ug);
      if (!prefetch_unit() || !same_view) view_offsets(d.in[0]);  // same patch and view: offsets still valid
    }
  };
  // FOLD: scale / shift of this thread's channel quad in the cursor's chunk, requested right after the cursor moves
  // and used a chunk later, when that chunk's inputs go to LDS
  f32x4 co_sc = {1.f, 1.f, 1.f, 1.f}, co_sh = {0.f, 0.f, 0.f, 0.f};
  float co_floor = 0.f;
  auto load_coeffs = [&]() {
    if constexpr (!FOLD) return;
    co_floor = v_relu ? 0.f : -__builtin_inff();
    if (v_scale != nullptr) {  // uniform
      co_sc = *reinterpret_cast<const f32x4*>(v_scale + p_c0 + cc);
      co_sh = *reinterpret_cast<const f32x4*>(v_shift + p_c0 + cc);
    } else {
      co_sc = f32x4{1.f, 1.f, 1.f, 1.f};
      co_sh = f32x4{0.f, 0.f, 0.f, 0.f};
    }
  };
  auto load_inputs = [&]() {
    if constexpr (LEAN) {  // slices are multiples of 8 channels: every chunk is full
      // (the resource is put together here, three scalar instructions: carried through the loop as a 128-bit value it
      // pulled the whole cursor into vector registers)
      const __amdgpu_buffer_rsrc_t v_rsrc =
          __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(v_ptr), 0, v_bytes, 0x00020000);
#pragma unroll
      for (int q = 0; q < IN_ITEMS; ++q) {
        const u32x4v v = __builtin_amdgcn_raw_buffer_load_b128(v_rsrc, static_cast<int>(voff[q]), p_c0 * 4, 0);
        reg_in[q] = __builtin_bit_cast(f32x4, v);
      }
      return;
    }
    pf_cnt = min(WKC, v_clen - p_c0);
#pragma unroll
    for (int q = 0; q < IN_ITEMS; ++q) {
      const unsigned off = voff[q] + static_cast<unsigned>(p_c0 + (cc < pf_cnt ? cc : 0));
      reg_in[q] = *reinterpret_cast<const f32x4*>(v_ptr + off);
    }
  };
  // The weight image of the cursor's chunk goes L2 -> LDS directly (LDS-DMA, 1 KB per wave and instruction, lane
  // linear like the image itself): no staging registers.
  auto dma_source = [&]() { return p_wimg + static_cast<long>(p_chunk) * WIMG; };  // the cursor's chunk image
  auto dma_weights = [&](float* w_dst, const float* wp, int q0 = 0, int q1 = 4) {    // W_ITEMS
#pragma unroll
    for (int q = q0; q < q1; ++q)
      __builtin_amdgcn_global_load_lds((gptr_t)(wp + (tid + q * kThreads) * 4),
                                       (lptr_t)(w_dst + (q * kThreads + wave * 64) * 4), 16, 0, 0);
  };
  auto store_chunk = [&](float* in_dst, int q0 = 0, int q1 = 3) {  // IN_ITEMS
    if constexpr (LEAN) {  // the dummy items of the last pass land behind the patch, inside the buffer (767 / 2 * WP < IN_FLOATS)
#pragma unroll
      for (int q = q0; q < q1; ++q) {
        const int it = tid + q * kThreads;
        f32x4 v = reg_in[q];
        if constexpr (FOLD) {  // affine, ReLU, and the zero padding back (pix < 0: outside the image)
          const unsigned inside = ~static_cast<unsigned>(pix[q] >> 31);
#pragma unroll
          for (int e = 0; e < 4; ++e)
            v[e] = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, fmaxf(fmaf(v[e], co_sc[e], co_sh[e]), co_floor)) & inside);
        }
        *reinterpret_cast<f32x2*>(&in_dst[(it >> 1) * WP + cc]) = f32x2{v[0], v[1]};
        *reinterpret_cast<f32x2*>(&in_dst[(it >> 1) * WP + cc + 2]) = f32x2{v[2], v[3]};
      }
      return;
    }
    const bool affine = v_scale != nullptr;  // BatchNorm apply + ReLU folded into the operand load
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (affine) {
      const int ch = p_c0 + (cc < pf_cnt ? cc : 0);
      sc = *reinterpret_cast<const f32x4*>(v_scale + ch);
      sh = *reinterpret_cast<const f32x4*>(v_shift + ch);
    }
#pragma unroll
    for (int q = 0; q < IN_ITEMS; ++q) {
      const int it = tid + q * kThreads;
      const bool keep = ((in_mask >> q) & 1u) && cc < pf_cnt;
      f32x4 v = reg_in[q];
      if (affine) {  // wave-uniform branches: plain views (most launches) only pay for the padding select
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
      }
      if (v_relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = keep ? v[e] : 0.f;  // zero padding AFTER the transform
      if (it < NPIX * 2) {  // 40-byte pixel rows: two 8-byte stores
        *reinterpret_cast<f32x2*>(&in_dst[(it >> 1) * WP + cc]) = f32x2{v[0], v[1]};
        *reinterpret_cast<f32x2*>(&in_dst[(it >> 1) * WP + cc + 2]) = f32x2{v[2], v[3]};
      }
    }
  };

#ifdef UNETPP_WINO_STAMPS
  unsigned long long st_acc[16] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = 0;
#endif
  // ---- epilogue of a finished unit: output transform, bias / ReLU / gate / accumulate, BatchNorm partial sums ----
  // The MFMAs take the weight fragment as their A operand and the transformed window as B, so register rr of acc[xi][nh]
  // of lane (t16, g) belongs to THIS lane's tile (16 * wave + t16, the tile whose windows it transforms) and to column
  // 16 * nh + 4 * g + rr: four consecutive output channels per accumulator, i.e. one 16-byte store per pixel and column
  // half straight from the registers (the 4 k-slot lanes of a tile write 64 contiguous bytes).  Rounds 1-5 had tiles in
  // the registers and one column per lane, and went through a 4 KB LDS transpose per wave (32 ds_write_b32 + 8
  // ds_read_b128 per unit, their waits, and a workgroup barrier before the next staging could reuse the scratch).
  auto epilogue = [&]() {
    const UnitGeom& ug = c_ug;
    const TileCols tc = decode_tile(a, ug.group);
    const unetpp_view& O = d.out[tc.ov];
    const bool interior = (ug.ty0 + TH <= d.H) && (ug.tx0 + TW <= d.W);
    const unsigned rs = static_cast<unsigned>(O.sy) * O.Ws * O.C, cs = static_cast<unsigned>(O.sx) * O.C;
    const unsigned tile_base = view_pixel_offset32(O, ug.n, ug.ty0, ug.tx0) + tc.nt * WNC;  // column 0 of the tile
    float* obase = O.ptr + tile_base;
    const float* gbase = O.gate != nullptr ? O.gate + tile_base : nullptr;
    const bool vec_out = ((O.C | O.c_off | tc.n_cnt) & 3) == 0 && (reinterpret_cast<uintptr_t>(O.ptr) & 15) == 0 &&
                         (O.gate == nullptr || (reinterpret_cast<uintptr_t>(O.gate) & 15) == 0);
    const bool want_stats = d.stats_partial != nullptr;
    const float floor_v = O.relu ? 0.f : -__builtin_inff();
    float s1[NH][4], s2[NH][4];
#pragma unroll
    for (int nh = 0; nh < NH; ++nh)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) s1[nh][rr] = s2[nh][rr] = 0.f;
    // bias folded into M[1][1]: its weight is +1 in all four outputs
    if (d.bias != nullptr && tc.n0 + WNC <= kBiasCols) {  // uniform; n0 is a multiple of 4 (fast_args)
#pragma unroll
      for (int nh = 0; nh < NH; ++nh) acc[5][nh] += *reinterpret_cast<const f32x4*>(&bias_lds[tc.n0 + 16 * nh + 4 * g]);
    } else if (d.bias != nullptr) {
#pragma unroll
      for (int nh = 0; nh < NH; ++nh)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const int col = 16 * nh + 4 * g + rr;
          acc[5][nh][rr] += col < tc.n_cnt ? d.bias[tc.n0 + col] : 0.f;
        }
    }
    WINO_STAMP(9);  // 9: epilogue: geometry, bias
    // this lane's tile inside the patch, and its first column
    const unsigned ty2 = 2 * (my_tile / TXN), tx2 = 2 * (my_tile % TXN);
    // 2x2 outputs of (nh, rr) for output row ap: y[0], y[1] = the two pixels of the row
    auto out_row = [&](int ap, int nh, int rr, float (&y)[2]) {
      float tb[4];
#pragma unroll
      for (int b = 0; b < 4; ++b)
        tb[b] = (ap == 0) ? (acc[b][nh][rr] + acc[4 + b][nh][rr]) + acc[8 + b][nh][rr]
                          : (acc[4 + b][nh][rr] - acc[8 + b][nh][rr]) - acc[12 + b][nh][rr];
      y[0] = fmaxf((tb[0] + tb[1]) + tb[2], floor_v);
      y[1] = fmaxf((tb[1] - tb[2]) - tb[3], floor_v);
    };
    if (interior && vec_out && tc.n_cnt == 16 * NH) {
      // ---- lean path (whole patch inside the image, all columns, 16-byte stores): no per-element predicates ----
      const bool has_gate = gbase != nullptr, acc_out = O.accumulate != 0;  // uniform
      const unsigned lane_off = ty2 * rs + tx2 * cs + 4 * g;
#pragma unroll
      for (int ap = 0; ap < 2; ++ap) {  // output row inside the 2x2 tile
        // The gate / previous-value reads of the row's stores are requested HERE, in front of the output transform:
        // vmcnt counts in order, so a read issued between two stores would wait for the store before it.
        unsigned off[2][NH];
        f32x4 gt[2][NH], old[2][NH];
#pragma unroll
        for (int bp = 0; bp < 2; ++bp)
#pragma unroll
          for (int nh = 0; nh < NH; ++nh) off[bp][nh] = lane_off + ap * rs + bp * cs + 16 * nh;
        if (has_gate) {
#pragma unroll
          for (int bp = 0; bp < 2; ++bp)
#pragma unroll
            for (int nh = 0; nh < NH; ++nh) gt[bp][nh] = *reinterpret_cast<const f32x4*>(gbase + off[bp][nh]);
        }
        if (acc_out) {
#pragma unroll
          for (int bp = 0; bp < 2; ++bp)
#pragma unroll
            for (int nh = 0; nh < NH; ++nh) old[bp][nh] = *reinterpret_cast<const f32x4*>(obase + off[bp][nh]);
        }
        // Output transform on column PAIRS (v_pk_add_f32 / v_pk_max_f32; same association order as out_row, so the same bits).
        f32x4 yv[2][NH];
        const f32x2 floor2 = {floor_v, floor_v};
#pragma unroll
        for (int nh = 0; nh < NH; ++nh) {
          f32x2 y[2][2];  // [pixel of the row][column pair]
#pragma unroll
          for (int cp = 0; cp < 2; ++cp) {
            f32x2 tb[4];
#pragma unroll
            for (int b = 0; b < 4; ++b) {
              const f32x2 m0 = cp ? acc[b][nh].hi : acc[b][nh].lo, m1 = cp ? acc[4 + b][nh].hi : acc[4 + b][nh].lo;
              const f32x2 m2 = cp ? acc[8 + b][nh].hi : acc[8 + b][nh].lo, m3 = cp ? acc[12 + b][nh].hi : acc[12 + b][nh].lo;
              tb[b] = (ap == 0) ? (m0 + m1) + m2 : (m1 - m2) - m3;
            }
            y[0][cp] = __builtin_elementwise_max((tb[0] + tb[1]) + tb[2], floor2);
            y[1][cp] = __builtin_elementwise_max((tb[1] - tb[2]) - tb[3], floor2);
          }
          yv[0][nh] = f32x4{y[0][0].x, y[0][0].y, y[0][1].x, y[0][1].y};
          yv[1][nh] = f32x4{y[1][0].x, y[1][0].y, y[1][1].x, y[1][1].y};
          if (want_stats) {
#pragma unroll
            for (int rr = 0; rr < 4; ++rr) {
              const float y0 = yv[0][nh][rr], y1 = yv[1][nh][rr];
              s1[nh][rr] += y0 + y1;
              s2[nh][rr] = fmaf(y0, y0, fmaf(y1, y1, s2[nh][rr]));
            }
          }
        }
        if (!has_gate && !acc_out) {  // plain stores on a path of their own: no load in it, so hipcc puts no vmcnt wait
#pragma unroll                        // between the stores
          for (int bp = 0; bp < 2; ++bp)
#pragma unroll
            for (int nh = 0; nh < NH; ++nh) {
              if constexpr (wino_exp::kNoOutStore) asm volatile("" ::"v"(yv[bp][nh]), "v"(off[bp][nh]));  // (ablation builds)
              else *reinterpret_cast<f32x4*>(obase + off[bp][nh]) = yv[bp][nh];
            }
          WINO_STAMP(10 + ap);  // 10, 11: epilogue rows
          continue;
        }
#pragma unroll
        for (int bp = 0; bp < 2; ++bp)
#pragma unroll
          for (int nh = 0; nh < NH; ++nh) {
            f32x4 v = yv[bp][nh];
            if (has_gate && !O.gate_sum) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (gt[bp][nh][e] > 0.f) ? v[e] : 0.f;
            }
            if (acc_out) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += old[bp][nh][e];
            }
            if (has_gate && O.gate_sum) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (gt[bp][nh][e] > 0.f) ? v[e] : 0.f;
            }
            *reinterpret_cast<f32x4*>(obase + off[bp][nh]) = v;
          }
        WINO_STAMP(10 + ap);
      }
    } else {
      // ---- general path (ragged patches, partial or unaligned column tiles): one predicated dword per value ----
#pragma unroll
      for (int ap = 0; ap < 2; ++ap) {
        const unsigned py = ty2 + ap;
#pragma unroll
        for (int nh = 0; nh < NH; ++nh)
#pragma unroll
          for (int rr = 0; rr < 4; ++rr) {
            float y[2];
            out_row(ap, nh, rr, y);
            const int col = 16 * nh + 4 * g + rr;
#pragma unroll
            for (int bp = 0; bp < 2; ++bp) {
              float v = y[bp];
              if ((col < tc.n_cnt) && (ug.ty0 + py < d.H) && (ug.tx0 + tx2 + bp < d.W)) {
                s1[nh][rr] += v;
                s2[nh][rr] = fmaf(v, v, s2[nh][rr]);
                const unsigned off = py * rs + (tx2 + bp) * cs + col;
                if (gbase != nullptr && !O.gate_sum) v = (gbase[off] > 0.f) ? v : 0.f;
                if (O.accumulate) v += obase[off];
                if (gbase != nullptr && O.gate_sum) v = (gbase[off] > 0.f) ? v : 0.f;
                obase[off] = v;
              }
            }
          }
      }
    }
    // (Round 6: starting the sums with C = 0 in the unit's first 32 MFMAs instead of these 128 v_mov was built and
    // measured -- profiles/r6/ab_wino_zero_in_mfma.txt.  As two C++ branches hipcc renames the group's accumulators and
    // spills; as one asm statement per group it is 1.4 % SLOWER on a step, because the opaque 8-MFMA blocks take the
    // staging code out from between the MFMAs, and the clearing it saves is worth 0.1-0.3 %: it sits in the epilogue.)
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
      for (int nh = 0; nh < NH; ++nh) acc[xi][nh] = f32x4{0.f, 0.f, 0.f, 0.f};
    if (want_stats) {
      // Sum over the wave's 16 tiles (the 16 lanes of a k-slot row: DPP adds in a fixed order, every lane ends with the
      // row's sum); per-wave sums go to LDS here, flush_stats() adds the four waves and writes the row behind a barrier.
#pragma unroll
      for (int nh = 0; nh < NH; ++nh)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          s1[nh][rr] = row16_sum(s1[nh][rr]);
          s2[nh][rr] = row16_sum(s2[nh][rr]);
          if (t16 == 0) {
            stat_lds[(wave * WNC + 16 * nh + 4 * g + rr) * 2 + 0] = s1[nh][rr];
            stat_lds[(wave * WNC + 16 * nh + 4 * g + rr) * 2 + 1] = s2[nh][rr];
          }
        }
    }
  };
  auto flush_stats = [&]() {
    if (d.stats_partial == nullptr || tid >= 16 * NH) return;
    const UnitGeom& ug = c_ug;
    const TileCols tc = decode_tile(a, ug.group);
    if (tid >= tc.n_cnt) return;
    float t1 = 0.f, t2 = 0.f;
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      t1 += stat_lds[(w * WNC + tid) * 2 + 0];
      t2 += stat_lds[(w * WNC + tid) * 2 + 1];
    }
    if constexpr (bn_fused) {  // column n0 + tid is always this thread's: a plain read-modify-write
      stat_lds[RUN0 + (tc.n0 + tid) * 2 + 0] += t1;
      stat_lds[RUN0 + (tc.n0 + tid) * 2 + 1] += t2;
      return;
    }
    float* dst = d.stats_partial + (ug.patch * a.Ncols + tc.n0 + tid) * 2;
    dst[0] = t1;
    dst[1] = t2;
  };

#ifdef UNETPP_WINO_STAMPS
  st_last = clock64();
#endif
  // ---- pipeline: at the top of chunk c the registers hold the inputs of chunk c+1 (loaded a whole chunk ago); they
  // go to the other LDS buffer, its weight image follows by DMA, the inputs of chunk c+2 are requested, and then the
  // 64 MFMAs of chunk c run without interruption.  One barrier per chunk. ----
  cache_view();
  prefetch_unit();
  view_offsets(d.in[0]);
  load_coeffs();
  load_inputs();
  dma_weights(w_tile, dma_source());
  store_chunk(in_tile);
  advance();
  load_coeffs();
  load_inputs();
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(IN_ITEMS) : "memory");  // the weight DMA has landed (the loads are younger)
  __syncthreads();
  WINO_STAMP(0);  // 0: prologue

  int c_chunk = 0, c_unit = 0;  // compute side: chunk of unit c_ug currently in LDS
  int cur = 0;                  // buffer being computed from
  while (true) {
    // ---- this chunk's first LDS operands are requested before the staging work below, which covers their latency
    const unsigned in_b = lds_offset(in_tile) + a_base * 4, w_b = lds_offset(w_tile) + b_base * 4;
    float dd[16], ddn[16], V[16];
    // LEAN: the window is read pairwise (ds_read2_b32: elements j, j + 1 of a row are WP floats apart) into register
    // pairs, one base register per window row, and transformed with packed adds: half the LDS and VALU instructions
    f32x2_t dp[8], dpn[8];
    unsigned row_b[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) row_b[i] = in_b + i * HWp * WP * 4;
    f32x4 us[2][2];  // two register sets of four weight fragments (steps 4g .. 4g+3), read one group ahead
    if constexpr (LEAN) {
      static_for<8>([&](auto ic) {
        constexpr int e = decltype(ic)::v, j = 2 * (e & 1);
        lds_read2_b32<j * WP, (j + 1) * WP>(dp[e], row_b[e >> 1]);
      });
    } else {
      static_for<16>([&](auto ic) {
        constexpr int e = decltype(ic)::v;
        lds_read_b32<((e / 4) * HWp + (e % 4)) * WP * 4>(dd[e], in_b);
      });
    }
    lds_read2st64_b64<0, 1>(us[0][0], w_b);
    lds_read2st64_b64<2, 3>(us[0][1], w_b);

    float* other = smem + (cur ^ 1) * BUF;
    if constexpr (!LEAN) {
      // The staging code is a few hundred sequential instructions; at the highest priority it gets through the issue
      // slots the co-resident wave's MFMA stream leaves and this wave is back at its own MFMAs sooner (+2 %).
      __builtin_amdgcn_s_setprio(3);
      store_chunk(other);
      dma_weights(other + IN_FLOATS, dma_source());
      advance();
      load_inputs();
    }
    WINO_STAMP(1);  // 1: staging store, cursor, load issue
    // LEAN: the staging of the next chunk is a hundred instructions; they are spread over the first three MFMA groups below,
    // where they issue in the shadow of this wave's own MFMAs.  Order: the LDS stores of the inputs requested most of a
    // chunk ago, then the weight DMA (hipcc makes any LDS store that follows a DMA wait for vmcnt(0)), the cursor, and
    // the requests for the chunk after next, which must stay the YOUNGEST vector-memory operations: the barrier waits
    // with vmcnt(IN_ITEMS), i.e. for the DMA but not for them.
    auto staging_piece = [&](auto gc) {
      constexpr int g = decltype(gc)::v;
      if constexpr (!LEAN) return;
      if constexpr (wino_exp::kNoStaging) {  // (ablation builds: wino_experiments.h)
        if constexpr (g == 2) advance();
        return;
      }
      if constexpr (g == 0 && !wino_exp::kNoStore) store_chunk(other);
      if constexpr (g == 0 && !wino_exp::kNoDma) dma_weights(other + IN_FLOATS, dma_source());
      if constexpr (g == 1) {
        advance();
        load_coeffs();
      }
      if constexpr (g == 2 && !wino_exp::kNoLoads) load_inputs();
    };

    // ---- per channel s the 4x4 window -> B^T d B in registers -> 32 MFMAs; every accumulator is touched once per
    // channel (no back-to-back dependence).  Raised wave priority for the MFMA phase: the co-resident workgroup's wave
    // on this SIMD is usually in its staging or epilogue VALU code, and the matrix pipe should never wait behind that.
    __builtin_amdgcn_s_setprio(2);
    if constexpr (LEAN) {
      lds_wait8x2(dp);
      lds_wait(us[0][0], us[0][1]);
      wino_input_transform_pk(dp, V);
    } else {
      lds_wait16(dd);
      lds_wait(us[0][0], us[0][1]);
      wino_input_transform(dd, V);
    }
    static_for<8>([&](auto gc) {  // group g = steps 4g .. 4g+3 of the chunk's 32 ([s][xi]); 8 MFMAs each
      constexpr int g = decltype(gc)::v, cs = g & 1, ns = cs ^ 1;
      if constexpr (g + 1 < 8) {
        lds_read2st64_b64<4 * (g + 1), 4 * (g + 1) + 1>(us[ns][0], w_b);
        lds_read2st64_b64<4 * (g + 1) + 2, 4 * (g + 1) + 3>(us[ns][1], w_b);
      }
      if constexpr (g == 0) {  // the second channel's window, needed from group 4 on
        if constexpr (LEAN) {
          static_for<8>([&](auto ic) {
            constexpr int e = decltype(ic)::v, j = 2 * (e & 1);
            lds_read2_b32<j * WP + 1, (j + 1) * WP + 1>(dpn[e], row_b[e >> 1]);
          });
        } else {
          static_for<16>([&](auto ic) {
            constexpr int e = decltype(ic)::v;
            lds_read_b32<(((e / 4) * HWp + (e % 4)) * WP + 1) * 4>(ddn[e], in_b);
          });
        }
      }
      // fences: the group's MFMAs (and the transform arithmetic feeding them) stay between the reads issued above
      // and the wait below -- left alone, the scheduler puts the reads straight in front of the wait again
      WINO_FENCE();
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // A = weight fragment, B = transformed window: the accumulator holds this lane's tile
        const int xi = (4 * g + q) & 15;
        const f32x4 up = us[cs][q >> 1];
        acc[xi][0] = __builtin_amdgcn_mfma_f32_16x16x4f32(up[(q & 1) * 2], V[xi], acc[xi][0], 0, 0, 0);
        if (NH == 2)
          acc[xi][NH - 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(up[(q & 1) * 2 + 1], V[xi], acc[xi][NH - 1], 0, 0, 0);
      }
      staging_piece(gc);
      WINO_FENCE();
      if constexpr (g == 0) {
        if constexpr (LEAN) lds_wait8x2(dpn);
        else lds_wait16(ddn);
      }
      if constexpr (g + 1 < 8) lds_wait(us[ns][0], us[ns][1]);
      if constexpr (g == 3) {
        WINO_STAMP(2);  // 2: first half of the MFMA phase
        if constexpr (LEAN) wino_input_transform_pk(dpn, V);
        else wino_input_transform(ddn, V);  // (inside group 4's fences it interleaves with the MFMAs, and is 1 % slower)
      }
    });
    __builtin_amdgcn_s_setprio(0);
    WINO_STAMP(4);  // 4: second half of the MFMA phase
    // Raw barrier: __syncthreads() carries a workgroup fence for which hipcc emits `s_waitcnt vmcnt(0) lgkmcnt(0)` in
    // front of the s_barrier of THIS loop (the epilogue's global stores of an earlier iteration may be outstanding): that
    // also drains the three input loads of the chunk after next, which were requested in MFMA group 2 of this very chunk
    // precisely so that they would have more than a chunk to arrive (and which the explicit vmcnt(IN_ITEMS) leaves in
    // flight).  What the barrier has to order is complete in every wave without it: its fragment reads of the current
    // buffers and its staging stores (lgkmcnt(0)), its share of the next weight image (vmcnt(IN_ITEMS)).
    // (round 5: 19.765 -> 19.715 ms per step against the __syncthreads() form, profiles/r5/ab_wino_raw_barrier.txt)
    if constexpr (!wino_exp::kNoBarrier) {
      if (wino_exp::kLaxWait && c_chunk == 0 && c_unit > 0)  // (ablation builds: first chunk behind an epilogue)
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(IN_ITEMS + 4 + 4 * NH) : "memory");
      else
      asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(IN_ITEMS) : "memory");
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
    }
    WINO_STAMP(5);  // 5: barrier
    if (c_chunk + 1 == a.n_chunks) {
      // Collect the prefetched inputs BEFORE the epilogue issues its stores: vmcnt counts in order, so a wait for
      // these loads placed after the stores (hipcc puts vmcnt(0) in front of the next staging store) would also wait
      // for the unit's stores to reach memory.  The loads are a whole MFMA phase old here.
#pragma unroll
      for (int q = 0; q < IN_ITEMS; ++q) asm volatile("" : "+v"(reg_in[q]));
      if constexpr (wino_exp::kNoEpilogue) {  // (ablation builds: the accumulators stay live, nothing else happens)
#pragma unroll
        for (int xi = 0; xi < 16; ++xi)
#pragma unroll
          for (int nh = 0; nh < NH; ++nh) {
            asm volatile("" : "+v"(acc[xi][nh]));
            acc[xi][nh] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
      } else {
        epilogue();  // stores drain while the next unit computes
      }
      WINO_STAMP(6);  // 6: epilogue
      if (d.stats_partial != nullptr) {  // uniform: the per-wave sums are in LDS; the next epilogue is a chunk barrier away
        __syncthreads();
        flush_stats();
      }
      WINO_STAMP(7);  // 7: barrier after the epilogue
      step_unit(c_ug);
      c_chunk = 0;
      if (++c_unit == n_units) break;
    } else {
      ++c_chunk;
    }
    cur ^= 1;
    in_tile = smem + cur * BUF;
    w_tile = in_tile + IN_FLOATS;
  }
#ifdef UNETPP_WINO_STAMPS
  if (lane == 0) {
    for (int i = 0; i < 16; ++i)
      if (i != 8) atomicAdd(&g_wino_stamps[i], st_acc[i]);
    atomicAdd(&g_wino_stamps[8], 1ull);
  }
#endif
  if constexpr (BNF) {
    __syncthreads();  // the last unit's sums are in
    bn_rows_store<kThreads>(d.stats_partial, a.Ncols, stat_lds + RUN0);
  }
}

}  // namespace

#ifdef UNETPP_WINO_STAMPS
extern "C" int unetpp_debug_wino_stamps(unsigned long long* out16, int reset) {  // profiling builds only
  if (out16 != nullptr && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_wino_stamps), sizeof(g_wino_stamps)) != hipSuccess)
    return UNETPP_ELAUNCH;
  if (reset) {
    const unsigned long long zero[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_wino_stamps), zero, sizeof(zero)) != hipSuccess) return UNETPP_ELAUNCH;
  }
  return UNETPP_OK;
}
#endif

bool wino_applies(const unetpp_gemm_desc* d) {
  return d != nullptr && d->taps == 9 && (d->flags & (UNETPP_GEMM_DIRECT | UNETPP_GEMM_BF16)) == 0;
}

int launch_gemm_wino(const unetpp_gemm_desc* d, hipStream_t st, long* bn_rows) {
  FastArgs a;
  if (!wino_applies(d) || !fast_args(d, a, WKC, WNC) || d->weight_image == nullptr) return UNETPP_EINVAL;
  if (d->stats_partial != nullptr && d->n_out != 1) return UNETPP_EINVAL;
  const int cus = device_cu_count();
  if (cus <= 0) return UNETPP_ELAUNCH;
  long workers = (2L * cus) & ~7L;  // persistent grid: two workgroups per CU (= the kernel's launch bounds)
#if defined(UNETPP_WINO_STAMPS) || defined(UNETPP_WINO_EXP)
  if (opt_value(OPT_WINO_ONE_PER_CU, 0) == 1) workers = cus & ~7L;  // waves alone on their SIMD
#endif
  if (workers < 8) workers = 8;
  if (workers > kBnFusedRows) workers = kBnFusedRows;
  const dim3 grid(static_cast<unsigned>(a.total_blocks <= workers ? a.total_blocks : workers)), block(kThreads);
  bool narrow = true;  // no output view wider than 16 channels: the second column half is never used
  for (int i = 0; i < d->n_out; ++i) narrow = narrow && d->out[i].c_len <= 16;
  // lean staging: launch-sized fp32 tensors, whole 8-channel chunks, 2 GB at most; mode 1 = no transform on load
  bool lean = opt_value(OPT_WINO_NO_LEAN, 0) == 0, fold = false;
  for (int i = 0; i < d->n_in; ++i) {
    const unetpp_view& v = d->in[i];
    lean = lean && (v.c_len & 7) == 0 && v.Hs == d->H && v.Ws == d->W && v.sy == 1 && v.sx == 1 && v.oy == 0 &&
           v.ox == 0 && static_cast<long>(d->N) * v.Hs * v.Ws * v.C * 4 <= 0x7fffffffL;
    fold = fold || v.scale != nullptr || v.relu != 0;
  }
  const int mode = !lean ? 0 : (fold ? 2 : 1);
  a.bn_in_kernel = (mode != 0 && bn_rows_per_workgroup(d, a.Ncols)) ? 1 : 0;  // (the general kernel: per-block rows)
#define UNETPP_LAUNCH_WINO_M(L, NHV)                                                                    \
  do {                                                                                                  \
    if (mode == 0) hipLaunchKernelGGL((gemm_wino_kernel<L, NHV, 0>), grid, block, 0, st, a);            \
    else if (a.bn_in_kernel && mode == 1) hipLaunchKernelGGL((gemm_wino_kernel<L, NHV, 1, true>), grid, block, 0, st, a); \
    else if (a.bn_in_kernel) hipLaunchKernelGGL((gemm_wino_kernel<L, NHV, 2, true>), grid, block, 0, st, a);             \
    else if (mode == 1) hipLaunchKernelGGL((gemm_wino_kernel<L, NHV, 1>), grid, block, 0, st, a);       \
    else hipLaunchKernelGGL((gemm_wino_kernel<L, NHV, 2>), grid, block, 0, st, a);                      \
  } while (0)
#define UNETPP_LAUNCH_WINO(L)                 \
  do {                                        \
    if (narrow) UNETPP_LAUNCH_WINO_M(L, 1);   \
    else UNETPP_LAUNCH_WINO_M(L, 2);          \
  } while (0)
  if (a.log2tw == 5) UNETPP_LAUNCH_WINO(5);
  else if (a.log2tw == 4) UNETPP_LAUNCH_WINO(4);
  else UNETPP_LAUNCH_WINO(3);
#undef UNETPP_LAUNCH_WINO
#undef UNETPP_LAUNCH_WINO_M
  note_kernel("gemm_wino_kernel");
  if (a.bn_in_kernel && bn_rows != nullptr) *bn_rows = grid.x;
  return launch_status();
}

}  // namespace unetpp

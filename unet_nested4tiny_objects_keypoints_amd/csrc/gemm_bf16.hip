// bf16-storage twin of the multi-view pixel GEMM (forward and input gradient of the 3x3 convolutions, the pointwise
// GEMMs of the 2x2 transposed convolution) for BASELINE configs[3]/[4]: activations in HBM are bf16 NHWC, the GEMM
// runs as a direct implicit GEMM on v_mfma_f32_32x32x16_bf16 with fp32 accumulators (no Winograd: its transforms
// need more than 8 mantissa bits), bias / BatchNorm coefficients / BatchNorm partial sums stay fp32.
//
// Same skeleton as gemm_fast.hip (persistent workgroups over (256-pixel patch, 32*NT-column) units, K in chunks, the
// next chunk's global loads issued into registers before the MFMA loop of the current one and written to LDS after
// it, LDS weight image per (column tile, chunk) packed once per pass) with the bf16 geometry:
//   * K chunk = 32 channels = 64 bytes per pixel: a 16-byte staging item carries 8 channels, four items per pixel,
//     LDS pixel stride 80 bytes (conflict-free ds_read_b128, every tap's address = lane base + immediate);
//   * MFMA operands: lane (j = lane & 31, h = lane >> 5) holds A[pixel j][k = 8h..8h+7] and B[k = 8h..8h+7][column j],
//     i.e. ONE 16-byte LDS read per operand and MFMA (32x32x16: 16 K per instruction, 32 cycles);
//   * weight image [tap][g 2][column 32][h 2][8 bf16]: the 64 lanes of a B read cover 1 KB contiguously;
//   * the BatchNorm-apply + ReLU load transform runs in fp32 between the global load and the LDS store and is
//     rounded to bf16 there (the MFMA takes bf16 operands); zero padding is applied after it;
//   * epilogue: bias, ReLU, rounding to bf16, BatchNorm partial sums OF THE ROUNDED VALUES (the statistics describe
//     the tensor that is stored), ReLU gate / accumulate / gate-of-the-sum read-modify-write in fp32; the accumulator
//     tile is transposed through LDS so that a lane stores 8 consecutive channels (16 bytes) of one pixel.
#include "bf16_common.h"
#include "common.h"
#include "gemm_units.h"

namespace unetpp {
namespace {

// In-kernel phase stamps (profiling builds only: -DUNETPP_BF16_STAMPS, tools/gbf_stamps.py): wave 0 of every workgroup adds
// the cycles it spent in each phase of its (unit, chunk) stream to a global table.
#ifdef UNETPP_BF16_STAMPS
__device__ unsigned long long g_gbf_stamps[16];
#define GBF_STAMP(i)                           \
  do {                                         \
    const unsigned long long now_ = clock64(); \
    st_acc[i] += now_ - st_last;               \
    st_last = now_;                            \
  } while (0)
#else
#define GBF_STAMP(i) \
  do {               \
  } while (0)
#endif

// Workgroups per CU = the register budget.  Round 5: TWO everywhere (256 registers, no spills, no scratch).  Until round 4
// the one-tile and the statistics instantiations ran three per CU at 168 registers with 3-23 registers spilled; since the
// LDS-DMA kernel took the plain launches (gemm_bf16_dma.hip) this kernel only runs the launches with a load transform --
// one per step in the benchmark configurations (conv2 of X_0,0 with the BatchNorm fold) -- and the A/B of round 4 had the
// spill-free two-per-CU build equal to the spilling three-per-CU one on it (profiles/r4, section 5 of DESIGN.md).
#ifndef UNETPP_BF16_WGS
#define UNETPP_BF16_WGS(TAPS, NT) 2
#endif
#ifndef UNETPP_BF16_STATS_WGS
#define UNETPP_BF16_STATS_WGS 2
#endif
constexpr int BKC = 32;        // channels per K chunk
constexpr int BPIX = 80;       // LDS bytes per staged pixel (64 + 16 pad)
constexpr int BSTEP = 1024;    // bytes of one (tap, g) weight step: 32 columns x 16 k x 2 B

// STATS: the launch takes BatchNorm partial sums (per-column sums over pixels).  Those launches keep the pixel on the
// MFMA row (accumulator register = pixel, lane = column: a column's sum is lane-local) and transpose the tile through
// LDS for 16-byte stores.  All other launches SWAP the MFMA operands (A = weights, B = pixels): the accumulator tile
// comes out as D[column][pixel], i.e. a lane holds 16 channels of ONE pixel in groups of four; one
// v_permlane32_swap per register pair pairs the groups of lanes (j, 0) and (j, 1) into 8 consecutive channels, and
// the tile leaves as two 16-byte stores per lane straight from registers -- no LDS round trip, no wave barriers.
template <int TAPS, int LOG2TW, int NT, bool STATS>
__global__ __launch_bounds__(kThreads, STATS ? UNETPP_BF16_STATS_WGS : UNETPP_BF16_WGS(TAPS, NT)) void gemm_bf16_kernel(const FastArgs a) {
  constexpr int HALO = (TAPS == 9) ? 1 : 0;
  constexpr int TW = 1 << LOG2TW, TH = kBlockPixels >> LOG2TW;
  constexpr int HWp = TW + 2 * HALO, HHp = TH + 2 * HALO;
  constexpr int NPIX = HWp * HHp;
  constexpr int MAXPIX = (TAPS == 9) ? kMaxHaloPixels : kBlockPixels;
  constexpr int IN_BYTES = MAXPIX * BPIX;  // 27200 / 20480: also >= the 4 x 4 KB transpose scratch of the epilogue
  constexpr int IMG = TAPS * 2 * BSTEP;    // bytes of one (column tile, chunk) image
  constexpr int IN_ITEMS = (NPIX * 4 + kThreads - 1) / kThreads;
  constexpr int W_ITEMS = (NT * IMG / 16 + kThreads - 1) / kThreads;
  static_assert(IN_BYTES >= 4 * 4096, "epilogue scratch does not fit");
  __shared__ __attribute__((aligned(16))) unsigned char smem[IN_BYTES + NT * IMG];
  unsigned char* in_tile = smem;
  unsigned char* w_tile = smem + IN_BYTES;
  __shared__ float stat_lds[4 * 32 * 2];  // the four waves' BatchNorm partial sums of a column tile

  const unetpp_gemm_desc& d = a.d;
  const int tid = threadIdx.x;
  const int wave = tid >> 6, lane = tid & 63, j = lane & 31, h = lane >> 5;

  // A workgroup walks a CONTIGUOUS run of units (column group fastest, then the patch along x, y, image): both cursors
  // (prefetch side, compute side) are decoded once and then stepped with scalar adds and compares -- at bf16 MFMA speed
  // the divisions and 64-bit multiplies of a per-unit decode (150 integer multiplies per unit, quarter rate) cost
  // more issue time than the unit's 36 MFMAs.
  // (The pointwise GEMMs of the transposed convolutions keep the round-robin order inside an XCD: their units are
  // short and write four strided phase views; contiguous runs made them 1.3x slower.)
  constexpr bool CONTIG = TAPS == 9;
  const UnitRange ur = CONTIG ? my_contiguous_unit_range(a.total_blocks) : my_unit_range(a.total_blocks);
  const long my_units = ur.count;
  if (my_units == 0) return;
  long p_index = ur.first, c_index = ur.first;  // round-robin mode: the cursors' unit indices
  auto step_unit = [&](UnitGeom& u, long& index) {
    if constexpr (!CONTIG) {
      index += ur.step;
      u = decode_unit<LOG2TW>(a, index);
      return;
    }
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

  int apix[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int p = 64 * wave + 32 * mt + j;
    apix[mt] = ((p >> LOG2TW) * HWp + (p & (TW - 1))) * BPIX + h * 16;
  }
  const int wb = (j * 2 + h) * 16;

  f32x16 acc[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][mt][r] = 0.f;

  // ---- prefetch side ----
  u32x4 reg_in[IN_ITEMS], reg_w[W_ITEMS];
  unsigned voff[IN_ITEMS];  // bf16 element offsets (< 2^31, fast_args)
  unsigned in_mask = 0;
  int pf_cnt = 0;
  long p_unit = 0;
  int p_s = 0, p_c0 = 0, p_chunk = 0;
  int p_n = 0, p_ty0 = 0, p_tx0 = 0;
  const unsigned char* p_wimg = nullptr;
  const unsigned char* wimg_base = reinterpret_cast<const unsigned char*>(d.weight_image);
  // the image a chunk needs is (column group, chunk): with ONE chunk per unit and ONE column group every unit of the
  // launch uses the same image, which then stays in LDS (short-K layers: K <= 32 into <= 32*NT columns)
  const bool w_resident = a.n_chunks == 1 && a.n_groups == 1;
  bool w_loaded = false;

  // Per-thread constants of the staging items: halo coordinates and validity (the patch shape is a template parameter).
  // Interior patches (all but the image border) need no clamping and no bounds mask: an item's offset is the patch
  // origin (wave uniform) plus (hy * row stride + hx * column stride) of the view -- a unit's MFMAs take 0.5 us per wave,
  // so the ~250 instructions of per-item clamping and index arithmetic per unit were a third of its instruction stream.
  int item_hy[IN_ITEMS], item_hx[IN_ITEMS];
  unsigned item_valid = 0;
#pragma unroll
  for (int q = 0; q < IN_ITEMS; ++q) {
    const int it = tid + q * kThreads;
    const int hp = min(it >> 2, NPIX - 1);
    item_hy[q] = hp / HWp;
    item_hx[q] = hp - item_hy[q] * HWp;
    if (it < NPIX * 4) item_valid |= 1u << q;
  }
  bool p_interior = false;
  auto prefetch_unit = [&]() {  // geometry of the unit under the prefetch cursor
    const UnitGeom& g = p_ug;
    p_n = g.n;
    p_ty0 = g.ty0;
    p_tx0 = g.tx0;
    p_wimg = wimg_base + static_cast<long>(g.group) * NT * a.n_chunks * IMG;
    p_interior = p_ty0 >= HALO && p_tx0 >= HALO && p_ty0 + TH + HALO <= d.H && p_tx0 + TW + HALO <= d.W;  // uniform
    if (p_interior) {
      in_mask = item_valid;
      return;
    }
    in_mask = 0;
#pragma unroll
    for (int q = 0; q < IN_ITEMS; ++q) {
      const int y = p_ty0 + item_hy[q] - HALO, x = p_tx0 + item_hx[q] - HALO;
      if (((item_valid >> q) & 1u) && y >= 0 && y < d.H && x >= 0 && x < d.W) in_mask |= 1u << q;
    }
  };
  auto view_offsets = [&](const unetpp_view& V) {  // every item loads from a valid address (clamped on border patches)
    if (p_interior) {
      const unsigned origin = view_pixel_offset32(V, p_n, p_ty0 - HALO, p_tx0 - HALO);
      const unsigned rs = static_cast<unsigned>(V.sy) * V.Ws * V.C, cs = static_cast<unsigned>(V.sx) * V.C;
#pragma unroll
      for (int q = 0; q < IN_ITEMS; ++q) voff[q] = origin + static_cast<unsigned>(item_hy[q]) * rs + static_cast<unsigned>(item_hx[q]) * cs;
      return;
    }
#pragma unroll
    for (int q = 0; q < IN_ITEMS; ++q) {
      const int yy = min(max(p_ty0 + item_hy[q] - HALO, 0), d.H - 1), xx = min(max(p_tx0 + item_hx[q] - HALO, 0), d.W - 1);
      voff[q] = static_cast<unsigned>(view_pixel_offset(V, p_n, yy, xx));
    }
  };
  auto load_chunk = [&]() {
    const unetpp_view& V = d.in[p_s];
    const bf16_t* vp = reinterpret_cast<const bf16_t*>(V.ptr);
    pf_cnt = min(BKC, V.c_len - p_c0);
#pragma unroll
    for (int q = 0; q < IN_ITEMS; ++q) {
      const int cc = ((tid + q * kThreads) & 3) << 3;
      const unsigned off = voff[q] + static_cast<unsigned>(p_c0 + (cc < pf_cnt ? cc : 0));
      reg_in[q] = *reinterpret_cast<const u32x4*>(vp + off);
    }
    if (!(w_resident && w_loaded)) {  // uniform branch
      const unsigned char* wp = p_wimg + static_cast<long>(p_chunk) * IMG;
#pragma unroll
      for (int q = 0; q < W_ITEMS; ++q) {
        const unsigned it = min(tid + q * kThreads, NT * IMG / 16 - 1);
        const unsigned t = it / (IMG / 16), r = it - t * (IMG / 16);
        reg_w[q] = *reinterpret_cast<const u32x4*>(wp + static_cast<long>(t) * a.n_chunks * IMG + r * 16u);
      }
    }
  };
  auto store_chunk = [&]() {
    const unetpp_view& V = d.in[p_s];
    const bool affine = V.scale != nullptr;
    const int cq = (tid & 3) << 3;  // all items of a thread share one channel octet
    float sc[8], sh[8];
    if (affine) {
      const int ch = p_c0 + (cq < pf_cnt ? cq : 0);
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        sc[e] = V.scale[ch + e];
        sh[e] = V.shift[ch + e];
      }
    }
#pragma unroll
    for (int q = 0; q < IN_ITEMS; ++q) {
      const int it = tid + q * kThreads;
      const int hp = it >> 2, q4 = it & 3;
      const bool keep = ((in_mask >> q) & 1u) && (q4 << 3) < pf_cnt;
      u32x4 v = reg_in[q];
      if (affine || V.relu) {  // fp32 transform, rounded back to bf16 for the MFMA
        float f[8];
        unpack8(v, f);
        if (affine) {
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = fmaf(f[e], sc[e], sh[e]);
        }
        if (V.relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) f[e] = fmaxf(f[e], 0.f);
        }
        v = pack8(f);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = keep ? v[e] : 0u;  // zero padding AFTER the transform
      if (it < NPIX * 4) *reinterpret_cast<u32x4*>(&in_tile[hp * BPIX + (q4 << 4)]) = v;
    }
    if (!(w_resident && w_loaded)) {
#pragma unroll
      for (int q = 0; q < W_ITEMS; ++q) {
        const int it = tid + q * kThreads;
        if (it < NT * IMG / 16) *reinterpret_cast<u32x4*>(&w_tile[it * 16]) = reg_w[q];
      }
      w_loaded = true;
    }
  };
  struct Frag {
    u32x4 b[NT], a0, a1;
  };
  auto read_frag = [&](int step) {  // step = tap * 2 + g: the 16 channels [16g, 16g + 16) of the chunk at one tap
    const int tap = step >> 1, g = step & 1;
    const int tpix = (TAPS == 9) ? ((tap / 3) * HWp + (tap % 3)) * BPIX : 0;
    Frag f;
#pragma unroll
    for (int t = 0; t < NT; ++t) f.b[t] = *reinterpret_cast<const u32x4*>(&w_tile[t * IMG + step * BSTEP + wb]);
    f.a0 = *reinterpret_cast<const u32x4*>(&in_tile[apix[0] + tpix + g * 32]);
    f.a1 = *reinterpret_cast<const u32x4*>(&in_tile[apix[1] + tpix + g * 32]);
    return f;
  };

  // Accumulator register r of lane (j, h): pixel 64*wave + 32*mt + 4h + c(r), c(r) = (r&3) + 8*(r>>2), column j.
  auto epilogue_stats = [&]() {
    const UnitGeom& g = c_ug;
    const bool interior = (g.ty0 + TH <= d.H) && (g.tx0 + TW <= d.W);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const TileCols tc = decode_tile(a, g.group * NT + t);
      const unetpp_view& O = d.out[tc.ov];
      bf16_t* optr = reinterpret_cast<bf16_t*>(O.ptr);
      const bf16_t* gptr = reinterpret_cast<const bf16_t*>(O.gate);
      const bool col_ok = j < tc.n_cnt;
      const float bj = (d.bias != nullptr && col_ok) ? d.bias[tc.n0 + j] : 0.f;
      const long row_stride = static_cast<long>(O.sy) * O.Ws * O.C, col_stride = static_cast<long>(O.sx) * O.C;
      const long tile_base = view_pixel_offset(O, g.n, g.ty0, g.tx0) + tc.nt * 32;  // column 0 of the tile
      float s1 = 0.f, s2sum = 0.f;
      float* scratch = reinterpret_cast<float*>(in_tile) + wave * 1024;  // [32 pixels][32 columns] fp32
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int prow = (64 * wave + 32 * mt) >> LOG2TW;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          const int dy = c >> LOG2TW, dx = c & (TW - 1);
          float v = acc[t][mt][r] + bj;
          if (O.relu) v = fmaxf(v, 0.f);
          v = bf_round(v);
          const bool ok = col_ok && (interior || ((g.ty0 + prow + dy < d.H) && (g.tx0 + 4 * h + dx < d.W)));
          if (ok) {
            s1 += v;
            s2sum = fmaf(v, v, s2sum);
          }
          scratch[(c + 4 * h) * 32 + (j ^ ((c & 3) << 3))] = v;  // column XOR by pixel: conflict-free 32-byte reads below
          acc[t][mt][r] = 0.f;
        }
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int pass = 0; pass < 2; ++pass) {
          const int pi = (lane >> 2) + 16 * pass, q8 = (lane & 3) << 3;  // pixel inside the MFMA tile, first column
          const int p = 64 * wave + 32 * mt + pi;
          const int py = p >> LOG2TW, px = p & (TW - 1);
          const int sw = ((pi & 3) << 3);
          const f32x4 v0 = *reinterpret_cast<const f32x4*>(&scratch[pi * 32 + (q8 ^ sw)]);
          const f32x4 v1 = *reinterpret_cast<const f32x4*>(&scratch[pi * 32 + ((q8 ^ sw) + 4)]);
          float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
          if (q8 < tc.n_cnt && (interior || ((g.ty0 + py < d.H) && (g.tx0 + px < d.W)))) {
            const long off = tile_base + py * row_stride + px * col_stride + q8;
            float gt[8];
            if (gptr != nullptr) unpack8(*reinterpret_cast<const u32x4*>(gptr + off), gt);
            if (gptr != nullptr && !O.gate_sum) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (gt[e] > 0.f) ? v[e] : 0.f;
            }
            if (O.accumulate) {
              float old[8];
              unpack8(*reinterpret_cast<const u32x4*>(optr + off), old);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += old[e];
            }
            if (gptr != nullptr && O.gate_sum) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (gt[e] > 0.f) ? v[e] : 0.f;
            }
            *reinterpret_cast<u32x4*>(optr + off) = pack8(v);
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
      if (d.stats_partial != nullptr) {
        s1 += __shfl_xor(s1, 32);
        s2sum += __shfl_xor(s2sum, 32);
        float* wsc = stat_lds;
        if (h == 0) {
          wsc[(wave * 32 + j) * 2 + 0] = s1;
          wsc[(wave * 32 + j) * 2 + 1] = s2sum;
        }
        __syncthreads();
        if (tid < tc.n_cnt) {
          float t1 = 0.f, t2 = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            t1 += wsc[(w * 32 + tid) * 2 + 0];
            t2 += wsc[(w * 32 + tid) * 2 + 1];
          }
          float* dst = d.stats_partial + (g.patch * a.Ncols + tc.n0 + tid) * 2;
          dst[0] = t1;
          dst[1] = t2;
        }
        __syncthreads();
      }
    }
  };

  int epi_py[2], epi_px[2];  // patch coordinates of this lane's pixel in the two MFMA pixel tiles (swapped epilogue)
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int p = 64 * wave + 32 * mt + j;
    epi_py[mt] = p >> LOG2TW;
    epi_px[mt] = p & (TW - 1);
  }
  // ---- register-direct epilogue (swapped operands).  Register r of acc[t][mt] of lane (j, h): output column
  // (r & 3) + 8 * (r >> 2) + 4 * h of pixel 64 * wave + 32 * mt + j. ----
  auto epilogue_direct = [&]() {
    const UnitGeom& g = c_ug;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const TileCols tc = decode_tile(a, g.group * NT + t);
      const unetpp_view& O = d.out[tc.ov];
      bf16_t* optr = reinterpret_cast<bf16_t*>(O.ptr);
      const bf16_t* gptr = reinterpret_cast<const bf16_t*>(O.gate);
      // 32-bit element offsets (fast_args: every tensor < 2^31 elements); the lane's pixel coordinates are constants
      const unsigned row_stride = static_cast<unsigned>(O.sy) * O.Ws * O.C, col_stride = static_cast<unsigned>(O.sx) * O.C;
      const unsigned tile_base = view_pixel_offset32(O, g.n, g.ty0, g.tx0) + tc.nt * 32;
      const bool rmw = gptr != nullptr || O.accumulate;  // uniform
      f32x4 b4[4];  // bias of this lane's four channel groups 8q + 4h .. + 3
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        b4[q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (d.bias != nullptr && 8 * q + 4 * h < tc.n_cnt) b4[q] = *reinterpret_cast<const f32x4*>(d.bias + tc.n0 + 8 * q + 4 * h);
      }
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int py = epi_py[mt], px = epi_px[mt];
        const bool pix_ok = (g.ty0 + py < d.H) && (g.tx0 + px < d.W);
        const unsigned pbase = tile_base + static_cast<unsigned>(py) * row_stride + static_cast<unsigned>(px) * col_stride;
        // Read-modify-write launches (ReLU gate and / or accumulation: the input gradients): the gate and previous values
        // of both 16-byte pieces of this pixel are requested here, in front of the packing arithmetic -- vmcnt counts
        // in order, so a read issued between two stores waits for the store in front of it.  Only in the two-tile 3x3
        // instantiation, which has the registers (256-register bound); the others spill with 16 more live registers.
        constexpr bool kHoist = TAPS == 9 && NT == 2;
        u32x4 gate_raw[2], old_raw[2];
        if constexpr (kHoist) {
          if (rmw) {
#pragma unroll
            for (int half = 0; half < 2; ++half) {
              const int c0 = 16 * half + 8 * (1 - h);
              const unsigned off = (pix_ok && c0 < tc.n_cnt) ? pbase + c0 : tile_base;  // dead pieces: any valid address
              if (gptr != nullptr) gate_raw[half] = *reinterpret_cast<const u32x4*>(gptr + off);
              if (O.accumulate) old_raw[half] = *reinterpret_cast<const u32x4*>(optr + off);
            }
          }
        }
        unsigned pk[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[t][mt][4 * q + e] + b4[q][e];
            if (O.relu) v[e] = fmaxf(v[e], 0.f);
            acc[t][mt][4 * q + e] = 0.f;
          }
          pk[q][0] = pack_bf2(v[0], v[1]);
          pk[q][1] = pack_bf2(v[2], v[3]);
        }
        // groups (1, 0) and (3, 2): afterwards lane (j, 0) holds columns 8..15 / 24..31, lane (j, 1) columns 0..7 / 16..23
#pragma unroll
        for (int half = 0; half < 2; ++half) {
          u32x4 out;
#pragma unroll
          for (int w2 = 0; w2 < 2; ++w2) {
            const auto r = __builtin_amdgcn_permlane32_swap(pk[2 * half + 1][w2], pk[2 * half][w2], false, false);
            out[w2] = r[0];
            out[2 + w2] = r[1];
          }
          const int c0 = 16 * half + 8 * (1 - h);  // first column of this lane's 8
          if (!rmw) {  // plain store: a path of its own, without loads -- hipcc puts a vmcnt(0) in front of a store whose
                       // value MAY come from a load, and with in-order vmcnt that waits for the previous store to land
            if (pix_ok && c0 < tc.n_cnt) *reinterpret_cast<u32x4*>(optr + pbase + c0) = out;
          } else if (pix_ok && c0 < tc.n_cnt) {
            const unsigned off = pbase + c0;
            if (gptr != nullptr || O.accumulate) {
              float v[8];
              unpack8(out, v);
              float gt[8];
              if (gptr != nullptr) unpack8(kHoist ? gate_raw[half] : *reinterpret_cast<const u32x4*>(gptr + off), gt);
              if (gptr != nullptr && !O.gate_sum) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (gt[e] > 0.f) ? v[e] : 0.f;
              }
              if (O.accumulate) {
                float old[8];
                unpack8(kHoist ? old_raw[half] : *reinterpret_cast<const u32x4*>(optr + off), old);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] += old[e];
              }
              if (gptr != nullptr && O.gate_sum) {
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (gt[e] > 0.f) ? v[e] : 0.f;
              }
              out = pack8(v);
            }
            *reinterpret_cast<u32x4*>(optr + off) = out;
          }
        }
      }
    }
  };

  prefetch_unit();
  view_offsets(d.in[0]);
  load_chunk();
  store_chunk();
  __syncthreads();

  long c_unit = 0;
  int c_chunk = 0;
#ifdef UNETPP_BF16_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = clock64();
  unsigned long long n_chunks_done = 0;
#endif
  while (true) {
    GBF_STAMP(0);  // 0: loop bookkeeping / prologue
    bool more = true;
    {
      int s2 = p_s, c2 = p_c0 + BKC;
      if (c2 >= d.in[p_s].c_len) {
        ++s2;
        c2 = 0;
      }
      if (p_chunk + 1 < a.n_chunks) {
        ++p_chunk;
        if (s2 != p_s) {
          p_s = s2;
          view_offsets(d.in[p_s]);
        }
        p_c0 = c2;
      } else if (p_unit + 1 < my_units) {
        ++p_unit;
        p_chunk = 0;
        p_s = 0;
        p_c0 = 0;
        step_unit(p_ug, p_index);
        prefetch_unit();
        view_offsets(d.in[0]);
      } else {
        more = false;
      }
    }
    load_chunk();  // unconditional (the cursor stays on the last chunk)
    GBF_STAMP(1);  // 1: cursor advance (unit decode, offsets) + load issue
    Frag cur = read_frag(0);
#ifdef UNETPP_BF16_EXP_NO_MFMA
    if (cur.a0[0] == 0x12345678u) acc[0][0][0] = 1.f;
#else
#pragma unroll
    for (int step = 0; step < TAPS * 2; ++step) {
      Frag nxt = cur;
      if (step + 1 < TAPS * 2) nxt = read_frag(step + 1);
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
        if constexpr (STATS) {
          acc[ct][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.a0),
                                                               __builtin_bit_cast(bf16x8, cur.b[ct]), acc[ct][0], 0, 0, 0);
          acc[ct][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.a1),
                                                               __builtin_bit_cast(bf16x8, cur.b[ct]), acc[ct][1], 0, 0, 0);
        } else {  // swapped roles: rows = output columns, columns = pixels (same fragments: see the lane maps)
          acc[ct][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.b[ct]),
                                                               __builtin_bit_cast(bf16x8, cur.a0), acc[ct][0], 0, 0, 0);
          acc[ct][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, cur.b[ct]),
                                                               __builtin_bit_cast(bf16x8, cur.a1), acc[ct][1], 0, 0, 0);
        }
      }
      cur = nxt;
    }
#endif
    GBF_STAMP(2);  // 2: LDS fragment reads + MFMAs
    __syncthreads();
    GBF_STAMP(3);  // 3: barrier after the MFMA loop
    if (c_chunk + 1 == a.n_chunks) {
      // Collect the prefetched chunk BEFORE the epilogue issues its stores: vmcnt counts in order, so the wait hipcc puts
      // in front of the next staging store would otherwise also wait for this unit's output to reach memory.
#pragma unroll
      for (int q = 0; q < IN_ITEMS; ++q) asm volatile("" : "+v"(reg_in[q]));
#pragma unroll
      for (int q = 0; q < W_ITEMS; ++q) asm volatile("" : "+v"(reg_w[q]));
      GBF_STAMP(4);  // 4: wait for the prefetched loads
#ifndef UNETPP_BF16_EXP_NO_EPILOGUE  // experiment builds only (tools/README.md): where does a unit's time go
      if constexpr (STATS) epilogue_stats();
      else epilogue_direct();
#else
#pragma unroll
      for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            if (acc[t][mt][r] == 123.456f) d.out[0].ptr[r] = 1.f;
            acc[t][mt][r] = 0.f;
          }
#endif
      if constexpr (STATS) __syncthreads();  // the transposing epilogue used the input tile as scratch
      GBF_STAMP(5);  // 5: epilogue
      step_unit(c_ug, c_index);
      ++c_unit;
      c_chunk = 0;
    } else {
      ++c_chunk;
    }
#ifdef UNETPP_BF16_STAMPS
    ++n_chunks_done;
#endif
    if (!more) break;
    store_chunk();
    GBF_STAMP(6);  // 6: staging stores (waits for the loads when no epilogue collected them)
    __syncthreads();
    GBF_STAMP(7);  // 7: barrier after the staging stores
  }
#ifdef UNETPP_BF16_STAMPS
  if (tid == 0) {
    for (int i = 0; i < 8; ++i) atomicAdd(&g_gbf_stamps[i], st_acc[i]);
    atomicAdd(&g_gbf_stamps[8], n_chunks_done);
    atomicAdd(&g_gbf_stamps[9], static_cast<unsigned long long>(my_units));
    atomicAdd(&g_gbf_stamps[10], 1ull);
  }
#endif
}

}  // namespace

#ifdef UNETPP_BF16_STAMPS
extern "C" int unetpp_debug_gbf_stamps(unsigned long long* out16, int reset) {  // profiling builds only
  if (out16 != nullptr && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_gbf_stamps), sizeof(g_gbf_stamps)) != hipSuccess)
    return UNETPP_ELAUNCH;
  if (reset) {
    const unsigned long long zero[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_gbf_stamps), zero, sizeof(zero)) != hipSuccess) return UNETPP_ELAUNCH;
  }
  return UNETPP_OK;
}
#endif

bool bf16_gemm_args(const unetpp_gemm_desc* d, FastArgs& a) {
  if (d == nullptr || (d->flags & UNETPP_GEMM_BF16) == 0) return false;
  if (!fast_args(d, a, BKC, 32)) return false;
  // 3x3 launches without a statistics epilogue feed TWO column tiles from one staged input patch when they can
  // (input gradients into several views, layers with >= 64 output channels)
  if (d->taps == 9 && d->stats_partial == nullptr && a.n_tiles % 2 == 0) {
    a.nt_unit = 2;
    a.n_groups = a.n_tiles / 2;
    a.total_blocks /= 2;
  }
  for (int i = 0; i < d->n_in; ++i)
    if (!bf16_view_aligned(d->in[i])) return false;
  for (int i = 0; i < d->n_out; ++i)
    if (!bf16_view_aligned(d->out[i])) return false;
  return true;
}

int launch_gemm_bf16(const unetpp_gemm_desc* d, hipStream_t st) {
  FastArgs a;
  if (!bf16_gemm_args(d, a) || d->weight_image == nullptr) return UNETPP_EINVAL;
  if (d->stats_partial != nullptr && d->n_out != 1) return UNETPP_EINVAL;
  const int cus = device_cu_count();
  if (cus <= 0) return UNETPP_ELAUNCH;
  // = the kernels' launch bounds (four workgroups per CU for the pointwise GEMMs: 128 VGPRs, spills, 2x slower)
  long workers = (static_cast<long>(UNETPP_BF16_WGS(d->taps, a.nt_unit)) * cus) & ~7L;
  if (d->stats_partial != nullptr) workers = (static_cast<long>(UNETPP_BF16_STATS_WGS) * cus) & ~7L;
  if (workers < 8) workers = 8;
  const dim3 grid(static_cast<unsigned>(a.total_blocks <= workers ? a.total_blocks : workers)), block(kThreads);
#define UNETPP_LAUNCH_BF16(T, NTU, ST)                                                                    \
  do {                                                                                                    \
    if (a.log2tw == 5) hipLaunchKernelGGL((gemm_bf16_kernel<T, 5, NTU, ST>), grid, block, 0, st, a);      \
    else if (a.log2tw == 4) hipLaunchKernelGGL((gemm_bf16_kernel<T, 4, NTU, ST>), grid, block, 0, st, a); \
    else hipLaunchKernelGGL((gemm_bf16_kernel<T, 3, NTU, ST>), grid, block, 0, st, a);                    \
  } while (0)
  const bool stats = d->stats_partial != nullptr;
  if (d->taps == 9) {
    if (stats) UNETPP_LAUNCH_BF16(9, 1, true);
    else if (a.nt_unit == 2) UNETPP_LAUNCH_BF16(9, 2, false);
    else UNETPP_LAUNCH_BF16(9, 1, false);
  } else if (a.nt_unit == 2) UNETPP_LAUNCH_BF16(1, 2, false);
  else UNETPP_LAUNCH_BF16(1, 1, false);
#undef UNETPP_LAUNCH_BF16
  note_kernel(d->taps == 9 ? "gemm_bf16_kernel<9>" : "gemm_bf16_kernel<1>");
  return launch_status();
}

}  // namespace unetpp

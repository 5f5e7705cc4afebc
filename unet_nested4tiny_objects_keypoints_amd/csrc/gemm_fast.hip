// Direct-summation fast path of the multi-view pixel GEMM (same math and MFMA maps as gemm_pix.hip, bit-identical
// results) for 16-byte aligned input views: the pointwise GEMMs of the 2x2 deconvolution / 1x1 convolution, and the
// 3x3 convolutions when UNETPP_GEMM_DIRECT forbids the Winograd kernel (gemm_wino.hip).
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
#include "gemm_units.h"

#ifndef UNETPP_FAST_PW_WGS   // workgroups per CU of the two-tile pointwise instantiation (A/B knob: tools/ab_lib.sh)
#define UNETPP_FAST_PW_WGS 3
#endif

namespace unetpp {
namespace {

constexpr int KC = 16;

// NT = column tiles per unit.  3x3 convolutions use NT = 1 (3 workgroups per CU); the pointwise GEMMs of the 2x2
// deconvolution (K = Cin only, N = 4*Cout) use NT = 2 so that one staged input patch feeds 64 columns.
template <int TAPS, int LOG2TW, int NT>
__global__ __launch_bounds__(kThreads, (TAPS == 1 ? (NT == 1 ? 4 : UNETPP_FAST_PW_WGS) : 3)) void gemm_fast_kernel(const FastArgs a) {
  constexpr int HALO = (TAPS == 9) ? 1 : 0;
  constexpr int TW = 1 << LOG2TW, TH = kBlockPixels >> LOG2TW;   // compile-time patch shape: every
  constexpr int HWp = TW + 2 * HALO, HHp = TH + 2 * HALO;        // division below is by a constant
  constexpr int NPIX = HWp * HHp;
  constexpr int MAXPIX = (TAPS == 9) ? kMaxHaloPixels : kBlockPixels;
  constexpr int KCP = 20;  // input pixel stride in LDS (floats)
  constexpr int IN_FLOATS = MAXPIX * KCP;
  constexpr int IMG = TAPS * 512;
  constexpr int IN_ITEMS = (NPIX * 4 + kThreads - 1) / kThreads;
  constexpr int W_ITEMS = (NT * IMG / 4 + kThreads - 1) / kThreads;
  __shared__ __attribute__((aligned(16))) float smem[IN_FLOATS + NT * IMG];
  float* in_tile = smem;
  float* w_tile = smem + IN_FLOATS;

  const unetpp_gemm_desc& d = a.d;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, j = lane & 31, h = lane >> 5;  // scalar: wave-uniform LDS addresses and tests stay on the scalar unit

  const UnitRange ur = my_unit_range(a.total_blocks);  // XCD-contiguous ranges, round-robin inside an XCD
  const long first_unit = ur.first, unit_step = ur.step, my_units = ur.count;
  if (my_units == 0) return;

  int apix[2];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int p = 64 * wave + 32 * mt + j;
    apix[mt] = (p >> LOG2TW) * HWp + (p & (TW - 1));
  }
  const int wb = j * 8 + ((h ^ ((j >> 3) & 1)) << 2);

  f32x16 acc[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][mt][r] = 0.f;

  // ---- prefetch side: the chunk that is loaded next.  Per-thread state: one 32-bit element offset per staging
  // item and one in-image bit per item; everything else is recomputed from tid where needed. ----
  f32x4 reg_in[IN_ITEMS], reg_w[W_ITEMS];
  unsigned voff[IN_ITEMS];   // element offsets (the launcher only takes this path for tensors < 2^31 elements)
  unsigned in_mask = 0;
  int pf_cnt = 0;            // valid channels of the chunk held in reg_in
  long p_unit = 0;           // index into this workgroup's units
  int p_s = 0, p_c0 = 0, p_chunk = 0;
  int p_n = 0, p_ty0 = 0, p_tx0 = 0;
  const float* p_wimg = nullptr;

  // (per-unit code: everything per-thread below derives from a thread index that is "produced" in place, so that hipcc
  // cannot hoist the items' halo coordinates out of the unit loop into registers the MFMA loop needs -- kept live across
  // the loop they were the 3-11 registers this kernel spilled under its 168 / 128-register bounds until round 4)
  auto in_place = [](int v) {
    asm volatile("" : "+v"(v));
    return v;
  };
  auto prefetch_unit = [&](long k) {  // geometry of unit k for the loads
    const UnitGeom g = decode_unit<LOG2TW>(a, first_unit + k * unit_step);
    p_n = g.n;
    p_ty0 = g.ty0;
    p_tx0 = g.tx0;
    p_wimg = d.weight_image + static_cast<long>(g.group) * NT * a.n_chunks * IMG;
    in_mask = 0;
    const int tidp = in_place(tid);
#pragma unroll
    for (int q = 0; q < IN_ITEMS; ++q) {
      const int it = tidp + q * kThreads;
      const int hp = it >> 2;
      const int hy = hp / HWp, hx = hp - hy * HWp;
      const int y = p_ty0 + hy - HALO, x = p_tx0 + hx - HALO;
      if ((it < NPIX * 4) && y >= 0 && y < d.H && x >= 0 && x < d.W) in_mask |= 1u << q;
    }
  };
  // The loads are straight-line code: every item loads from a VALID address (out-of-image halo pixels and padding
  // items are clamped into the image, channels past the view to channel 0) and the zeroing happens at the LDS
  // write.  Conditional loads would sit under divergent branches, where hipcc drains vmcnt at every join.
  auto view_offsets = [&](const unetpp_view& V) {
    const int tidp = in_place(tid);
#pragma unroll
    for (int q = 0; q < IN_ITEMS; ++q) {
      const int hp = min((tidp + q * kThreads) >> 2, NPIX - 1);
      const int hy = hp / HWp, hx = hp - hy * HWp;
      const int yy = min(max(p_ty0 + hy - HALO, 0), d.H - 1), xx = min(max(p_tx0 + hx - HALO, 0), d.W - 1);
      voff[q] = static_cast<unsigned>(view_pixel_offset(V, p_n, yy, xx));
    }
  };
  auto load_chunk = [&]() {
    const unetpp_view& V = d.in[p_s];
    pf_cnt = min(KC, V.c_len - p_c0);
#pragma unroll
    for (int q = 0; q < IN_ITEMS; ++q) {
      const int cc = ((tid + q * kThreads) & 3) << 2;
      const unsigned off = voff[q] + static_cast<unsigned>(p_c0 + (cc < pf_cnt ? cc : 0));
      reg_in[q] = *reinterpret_cast<const f32x4*>(V.ptr + off);
    }
    const float* wp = p_wimg + static_cast<long>(p_chunk) * IMG;
#pragma unroll
    for (int q = 0; q < W_ITEMS; ++q) {
      const unsigned it = min(tid + q * kThreads, NT * IMG / 4 - 1);
      const unsigned t = it / (IMG / 4), r = it - t * (IMG / 4);  // image of column tile t, float4 r
      reg_w[q] = *reinterpret_cast<const f32x4*>(wp + static_cast<long>(t) * a.n_chunks * IMG + r * 4u);
    }
  };
  auto store_chunk = [&]() {
    // optional load transform of the view (BatchNorm apply + ReLU folded into the consumer: the normalised
    // activation is never materialised).  All items of a thread share one channel quad, so the coefficients are
    // one float4 pair per chunk; zero padding stays zero because it is applied after the transform.
    const unetpp_view& V = d.in[p_s];
    const bool affine = V.scale != nullptr;
    f32x4 sc = {1.f, 1.f, 1.f, 1.f}, sh = {0.f, 0.f, 0.f, 0.f};
    if (affine) {
      const int ch = p_c0 + (((tid & 3) << 2) < pf_cnt ? ((tid & 3) << 2) : 0);
      sc = *reinterpret_cast<const f32x4*>(V.scale + ch);
      sh = *reinterpret_cast<const f32x4*>(V.shift + ch);
    }
#pragma unroll
    for (int q = 0; q < IN_ITEMS; ++q) {
      const int it = tid + q * kThreads;
      const int hp = it >> 2, q4 = it & 3;
      const bool keep = ((in_mask >> q) & 1u) && (q4 << 2) < pf_cnt;
      f32x4 v = reg_in[q];
      if (affine) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaf(v[e], sc[e], sh[e]);
      }
      if (V.relu) {
#pragma unroll
        for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
      }
#pragma unroll
      for (int e = 0; e < 4; ++e) v[e] = keep ? v[e] : 0.f;
      if (it < NPIX * 4) *reinterpret_cast<f32x4*>(&in_tile[hp * KCP + (q4 << 2)]) = v;
    }
#pragma unroll
    for (int q = 0; q < W_ITEMS; ++q) {
      const int it = tid + q * kThreads;
      if (it < NT * IMG / 4) *reinterpret_cast<f32x4*>(&w_tile[it * 4]) = reg_w[q];
    }
  };
  // one (tap, 8-channel group) step of LDS fragments: B for the 32 columns, A for both pixel tiles
  struct Frag {
    f32x4 b[NT], a0, a1;
  };
  auto read_frag = [&](int step) {
    const int tap = step >> 1, g = step & 1;
    const int tpix = (TAPS == 9) ? (tap / 3) * HWp + (tap % 3) : 0;
    Frag f;
#pragma unroll
    for (int t = 0; t < NT; ++t) f.b[t] = *reinterpret_cast<const f32x4*>(&w_tile[t * IMG + step * 256 + wb]);
    f.a0 = *reinterpret_cast<const f32x4*>(&in_tile[(apix[0] + tpix) * KCP + ((2 * g + h) << 2)]);
    f.a1 = *reinterpret_cast<const f32x4*>(&in_tile[(apix[1] + tpix) * KCP + ((2 * g + h) << 2)]);
    return f;
  };
  // epilogue of the unit whose K loop just finished: bias, ReLU, gate, store / accumulate, optional BatchNorm
  // partial sums.  Accumulator register r of lane (j, h) is pixel p = 64*wave + 32*mt + 4h + c(r),
  // c(r) = (r&3) + 8*(r>>2); with the compile-time patch shape its offset is
  // lane_base + (c >> LOG2TW)*row_stride + (c & (TW-1))*col_stride.
  auto epilogue = [&](long k) {
    const UnitGeom g = decode_unit<LOG2TW>(a, first_unit + k * unit_step);
    const bool interior = (g.ty0 + TH <= d.H) && (g.tx0 + TW <= d.W);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
    const TileCols tc = decode_tile(a, g.group * NT + t);
    const unetpp_view& O = d.out[tc.ov];
    const bool col_ok = j < tc.n_cnt;
    const float bj = (d.bias != nullptr && col_ok) ? d.bias[tc.n0 + j] : 0.f;
    const long row_stride = static_cast<long>(O.sy) * O.Ws * O.C, col_stride = static_cast<long>(O.sx) * O.C;
    const long tile_base = view_pixel_offset(O, g.n, g.ty0, g.tx0) + tc.nt * 32 + j;
    float s1 = 0.f, s2sum = 0.f;
    // 16-byte path: the accumulator tile (lane = one column, 16 pixels) is transposed through a per-wave 4 KB LDS
    // scratch so that a lane owns 4 consecutive channels of one pixel and the tile leaves as 4 dwordx4 stores of
    // whole 128-byte pixel rows instead of 16 dword stores (a dword-per-lane store tail is issue-bound: the 32
    // stores of a unit cost about as much as a K chunk of MFMAs).
    const bool vec_out = ((O.C | O.c_off | tc.n_cnt) & 3) == 0 && (reinterpret_cast<uintptr_t>(O.ptr) & 15) == 0 &&
                         (O.gate == nullptr || (reinterpret_cast<uintptr_t>(O.gate) & 15) == 0);
    if (vec_out) {
      float* scratch = in_tile + wave * 1024;  // [32 pixels][32 columns]; in_tile is free between the barriers
      const long tile_base4 = tile_base - j;   // column 0 of the tile
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int prow = (64 * wave + 32 * mt) >> LOG2TW;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int c = (r & 3) + 8 * (r >> 2);
          const int dy = c >> LOG2TW, dx = c & (TW - 1);
          float v = acc[t][mt][r] + bj;
          if (O.relu) v = fmaxf(v, 0.f);
          const bool ok = col_ok && (interior || ((g.ty0 + prow + dy < d.H) && (g.tx0 + 4 * h + dx < d.W)));
          if (ok) {
            s1 += v;
            s2sum = fmaf(v, v, s2sum);
          }
          scratch[(c + 4 * h) * 32 + j] = v;
          acc[t][mt][r] = 0.f;
        }
        __builtin_amdgcn_wave_barrier();
        // The four 16-byte pieces of this lane: offsets and validity first, then (read-modify-write launches only) all
        // gate / previous-value reads in one batch, then the stores.  vmcnt counts in order: a read issued between two
        // stores waits for the store before it, and hipcc guards every store that shares a path with a load by a
        // counted wait -- plain launches therefore take a path of their own with nothing but LDS reads and stores.
        const int q4 = (lane & 7) << 2;  // first column of the piece
        unsigned offs[4];  // elements: the fast kernels only take tensors below 2^31 elements (fast_args)
        bool live[4];
#pragma unroll
        for (int pass = 0; pass < 4; ++pass) {
          const int pi = (lane >> 3) + 8 * pass;  // pixel inside the MFMA tile
          const int p = 64 * wave + 32 * mt + pi;
          const int py = p >> LOG2TW, px = p & (TW - 1);
          live[pass] = q4 < tc.n_cnt && (interior || ((g.ty0 + py < d.H) && (g.tx0 + px < d.W)));
          offs[pass] = static_cast<unsigned>(tile_base4 + py * row_stride + px * col_stride + q4);
        }
        const bool has_gate = O.gate != nullptr, acc_out = O.accumulate != 0;  // uniform
        if (!has_gate && !acc_out) {
#pragma unroll
          for (int pass = 0; pass < 4; ++pass) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(&scratch[((lane >> 3) + 8 * pass) * 32 + q4]);
            if (live[pass]) *reinterpret_cast<f32x4*>(O.ptr + offs[pass]) = v;
          }
        } else {
#pragma unroll
          for (int half = 0; half < 2; ++half) {  // two pieces per batch: four spill under the 168-register bound (re-checked in round 5)
            f32x4 gt[2], old[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {  // dead pieces read the tile's first (always valid) address
              const unsigned lo = live[2 * half + u] ? offs[2 * half + u] : static_cast<unsigned>(tile_base4);
              if (has_gate) gt[u] = *reinterpret_cast<const f32x4*>(O.gate + lo);
              if (acc_out) old[u] = *reinterpret_cast<const f32x4*>(O.ptr + lo);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
              const int pass = 2 * half + u;
              f32x4 v = *reinterpret_cast<const f32x4*>(&scratch[((lane >> 3) + 8 * pass) * 32 + q4]);
              if (has_gate && !O.gate_sum) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (gt[u][e] > 0.f) ? v[e] : 0.f;
              }
              if (acc_out) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] += old[u][e];
              }
              if (has_gate && O.gate_sum) {
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = (gt[u][e] > 0.f) ? v[e] : 0.f;
              }
              if (live[pass]) *reinterpret_cast<f32x4*>(O.ptr + offs[pass]) = v;
            }
          }
        }
        __builtin_amdgcn_wave_barrier();
      }
    } else {
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
      const int prow = (64 * wave + 32 * mt) >> LOG2TW;  // first patch row of this MFMA tile
      const long lane_base = tile_base + prow * row_stride + (4 * h) * col_stride;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int c = (r & 3) + 8 * (r >> 2);
        const int dy = c >> LOG2TW, dx = c & (TW - 1);
        const bool ok = col_ok && (interior || ((g.ty0 + prow + dy < d.H) && (g.tx0 + 4 * h + dx < d.W)));
        if (ok) {
          float v = acc[t][mt][r] + bj;
          if (O.relu) v = fmaxf(v, 0.f);
          s1 += v;
          s2sum = fmaf(v, v, s2sum);
          const long off = lane_base + dy * row_stride + dx * col_stride;
          if (O.gate != nullptr && !O.gate_sum) v = (O.gate[off] > 0.f) ? v : 0.f;
          if (O.accumulate) v += O.ptr[off];
          if (O.gate != nullptr && O.gate_sum) v = (O.gate[off] > 0.f) ? v : 0.f;
          O.ptr[off] = v;
        }
        acc[t][mt][r] = 0.f;
      }
    }
    }
    if (d.stats_partial != nullptr) {  // the LDS tiles are free here (barrier after the MFMA loop)
      s1 += __shfl_xor(s1, 32);
      s2sum += __shfl_xor(s2sum, 32);
      if (h == 0) {
        w_tile[(wave * 32 + j) * 2 + 0] = s1;  // weight tile as scratch: the input tile may still be another
        w_tile[(wave * 32 + j) * 2 + 1] = s2sum;  // wave's transpose scratch
      }
      __syncthreads();
      if (tid < tc.n_cnt) {
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
          t1 += w_tile[(w * 32 + tid) * 2 + 0];
          t2 += w_tile[(w * 32 + tid) * 2 + 1];
        }
        float* dst = d.stats_partial + (g.patch * a.Ncols + tc.n0 + tid) * 2;
        dst[0] = t1;
        dst[1] = t2;
      }
      __syncthreads();  // before the next chunk overwrites the scratch
    }
    }  // column tiles of the unit
  };

  prefetch_unit(0);
  view_offsets(d.in[0]);
  load_chunk();
  store_chunk();
  __syncthreads();

  long c_unit = 0;   // compute side: unit and chunk currently in LDS
  int c_chunk = 0;
  while (true) {
    // ---- advance the prefetch cursor: next chunk of this unit, or chunk 0 of the next unit ----
    bool more = true;
    {
      int s2 = p_s, c2 = p_c0 + KC;
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
        prefetch_unit(p_unit);
        view_offsets(d.in[0]);
      } else {
        more = false;
      }
    }
    load_chunk();  // unconditional (the cursor stays on the last chunk): a load under a branch drains vmcnt at the join
    // ---- LDS -> MFMA for the current chunk: the fragments of step k+1 are read while the 8 MFMAs of step k
    // issue (both 8-channel groups always run; a short last chunk is zero-padded in LDS) ----
    Frag cur = read_frag(0);
#pragma unroll
    for (int step = 0; step < TAPS * 2; ++step) {
      Frag nxt = cur;
      if (step + 1 < TAPS * 2) nxt = read_frag(step + 1);
#pragma unroll
      for (int ct = 0; ct < NT; ++ct) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[ct][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a0[t], cur.b[ct][t], acc[ct][0], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < 4; ++t)
          acc[ct][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(cur.a1[t], cur.b[ct][t], acc[ct][1], 0, 0, 0);
      }
      cur = nxt;
    }
    __syncthreads();
    if (c_chunk + 1 == a.n_chunks) {
      epilogue(c_unit);  // stores drain while the next unit computes; the next chunk's loads are already in flight
      __syncthreads();   // the epilogue used the input tile as transpose scratch
      ++c_unit;
      c_chunk = 0;
    } else {
      ++c_chunk;
    }
    if (!more) break;
    store_chunk();
    __syncthreads();
  }
}

}  // namespace

int launch_gemm_fast(const unetpp_gemm_desc* d, hipStream_t st) {
  FastArgs a;
  if (!fast_args(d, a, KC) || d->weight_image == nullptr) return UNETPP_EINVAL;
  if (d->stats_partial != nullptr && d->n_out != 1) return UNETPP_EINVAL;
  if (d->taps == 1) {  // plain aligned pointwise launches whose weights fit LDS: gemm_pw.hip
    const int pw = launch_gemm_pw(d, a, st);
    if (pw != 1) return pw;
  }
  // persistent grid: at most 3 workgroups per CU (the kernel's LDS/VGPR budget), a multiple of 8
  const int cus = device_cu_count();
  if (cus <= 0) return UNETPP_ELAUNCH;
  const long per_cu = (d->taps == 1 && a.nt_unit == 1) ? 4 : (d->taps == 1 ? UNETPP_FAST_PW_WGS : 3);  // = the kernel's launch bounds
  long workers = (per_cu * cus) & ~7L;
  if (workers < 8) workers = 8;
  const dim3 grid(static_cast<unsigned>(a.total_blocks <= workers ? a.total_blocks : workers)), block(kThreads);
#define UNETPP_LAUNCH_FAST(T, NTU)                                                                      \
  do {                                                                                                  \
    if (a.log2tw == 5) hipLaunchKernelGGL((gemm_fast_kernel<T, 5, NTU>), grid, block, 0, st, a);        \
    else if (a.log2tw == 4) hipLaunchKernelGGL((gemm_fast_kernel<T, 4, NTU>), grid, block, 0, st, a);   \
    else hipLaunchKernelGGL((gemm_fast_kernel<T, 3, NTU>), grid, block, 0, st, a);                      \
  } while (0)
  if (d->taps == 9) UNETPP_LAUNCH_FAST(9, 1);
  else if (a.nt_unit == 2) UNETPP_LAUNCH_FAST(1, 2);
  else UNETPP_LAUNCH_FAST(1, 1);
#undef UNETPP_LAUNCH_FAST
  note_kernel(d->taps == 9 ? "gemm_fast_kernel<9>" : "gemm_fast_kernel<1>");
  return launch_status();
}

}  // namespace unetpp

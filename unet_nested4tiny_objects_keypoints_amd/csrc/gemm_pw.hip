// Pointwise (taps = 1) fp32 GEMM without an LDS stage for the activations: the 2x2 stride-2 transposed convolution
// (forward: one input view, four phase output views; input gradient: four phase input views, one output view with
// ReLU gate / accumulation) and the 1x1 convolution of the bilinear up path (/root/reference/models/unet.py:187,191).
//
// These launches move 768 bytes per 16 kFLOP (level 0: K = 64, N = 128): with the fp32 matrix pipe they sit at the HBM
// roofline, not at the MFMA one, and gemm_fast_kernel<1> ran them at 2.7 TB/s -- its time was the SUM of its memory and
// matrix phases (LDS staging of every 16-channel chunk, two to three workgroup barriers per chunk, an LDS transpose in
// the epilogue).  Here:
//   * the weight matrix sits in LDS for the kernel's lifetime ([K][N] fp32, at most 144 KB) as the A operand of
//     v_mfma_f32_16x16x4_f32: MFMA rows = output columns, fragment reads are lane-linear ds_read_b128;
//   * the activations are the B operand and never touch LDS: lane (pixel t16, k-slot g) loads 16 bytes = channels
//     16 q + 4 g .. + 3 of its pixel straight into the registers the MFMAs read (the four k-slot lanes of a pixel
//     fetch 64 contiguous bytes; the K order inside a 16-channel group is permuted the same way in the weight image);
//   * the accumulator of a 16 x 16 block then holds four consecutive output channels of the lane's own pixel: bias,
//     ReLU, gate, accumulate and the 16-byte stores happen in registers (non-temporal for outputs far beyond the L2);
//   * a wave owns 16 consecutive pixels of an image row and all N columns (in passes of at most 128); there is no
//     barrier after the weights have been staged, and 16 waves per CU hide the memory latency.
// Measured on the level-0 transposed convolution of BASELINE configs[1] (tools/probes/pw_direct_probe.hip): 80 us
// against 149 us.  The weights are taken from the image gemm_fast.hip uses (weight_image.hip, kind "fast": [column tile
// 32][chunk 16][g 2][col 32][half' 2][4]) and re-ordered while they are staged, so the pack plan does not change.
#include "common.h"
#include "gemm_units.h"
#include "lds_asm.h"

namespace unetpp {
namespace {

struct PwArgs {
  unetpp_gemm_desc d;
  int K, N;
  int tiles_x;      // 16-pixel tiles per image row (W / 16)
  int tiles_shift;  // log2(tiles_x) when it is a power of two, else -1
  int n_pass, n_kchunk;
  int nt_store;     // non-temporal stores
  long n_tiles;     // N_img * H * tiles_x  (< 2^31)
};

// QC = 16-channel groups per K chunk (the B fragments of a chunk live in 4 QC registers), NCBP = 16-column blocks per
// column pass (4 NCBP accumulator registers); EPI 0: every output view is a plain store (bias only), 1: ReLU / gate /
// accumulate per view.  Launched with 256, 512 or 1024 threads by the size of the weight image.
template <int QC, int NCBP, int EPI>
__global__ __launch_bounds__(1024, 1) void gemm_pw_kernel(const PwArgs a) {
  extern __shared__ __attribute__((aligned(16))) float pw_lds[];
  float* w_lds = pw_lds;
  float* b_lds = pw_lds + a.K * a.N;
  const unetpp_gemm_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, t16 = lane & 15, g = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), n_waves = blockDim.x >> 6;

  // ---- weights -> LDS, once: block (pass, chunk) holds [q QC][cb NCBP][g 4][t16 16][e 4] =
  // W[k = 16 (chunk QC + q) + 4 g + e][n = 16 (pass NCBP + cb) + t16]: every fragment read of the loop below is the
  // block's base plus a compile-time offset ----
  {
    const int NCB = a.N >> 4, NQ = a.K >> 4;
    for (int i = tid; i < NQ * NCB * 64; i += blockDim.x) {
      const int l = i & 63, blk = i >> 6;
      const int cbg = blk % NCB, qg = blk / NCB;
      const int g4 = l >> 4, tt = l & 15;
      const int nt = cbg >> 1, col = 16 * (cbg & 1) + tt, gg = g4 >> 1, h = g4 & 1;
      const f32x4 v = *reinterpret_cast<const f32x4*>(d.weight_image + (static_cast<long>(nt) * NQ + qg) * 512 +
                                                      ((gg * 32 + col) * 2 + (h ^ ((col >> 3) & 1))) * 4);
      const int pass = cbg / NCBP, cb = cbg - pass * NCBP, chunk = qg / QC, q = qg - chunk * QC;
      *reinterpret_cast<f32x4*>(w_lds + ((((pass * a.n_kchunk + chunk) * QC + q) * NCBP + cb) * 64 + l) * 4) = v;
    }
    for (int i = tid; i < a.N; i += blockDim.x) b_lds[i] = d.bias != nullptr ? d.bias[i] : 0.f;
  }
  __syncthreads();

  const unsigned wa = lds_offset(w_lds) + lane * 16;
  const long stride = static_cast<long>(gridDim.x) * n_waves;
  for (long tl = static_cast<long>(blockIdx.x) * n_waves + wave; tl < a.n_tiles; tl += stride) {
    const unsigned t = static_cast<unsigned>(tl);
    const unsigned row = a.tiles_shift >= 0 ? t >> a.tiles_shift : t / static_cast<unsigned>(a.tiles_x);
    const int x = static_cast<int>(t - row * a.tiles_x) * 16 + t16;
    const int n = static_cast<int>(row / static_cast<unsigned>(d.H)), y = static_cast<int>(row - n * d.H);

    // Cursors over the views: a view's fields (scalar loads) and this lane's pixel offset in it are only touched when
    // the cursor enters the view -- per 16 channels they cost more than the MFMAs they feed.
    int ov = 0, och = 0;  // output: view and channel of the next 16-column block
    float* optr = nullptr;
    const float* ogate = nullptr;
    int oclen = 0;
    bool orelu = false, oacc = false, ogsum = false;
    auto enter_out = [&]() {
      const unetpp_view& O = d.out[ov];
      const unsigned off = view_pixel_offset32(O, n, y, x) + 4 * g;
      optr = O.ptr + off;
      oclen = O.c_len;
      if constexpr (EPI != 0) {
        ogate = O.gate != nullptr ? O.gate + off : nullptr;
        orelu = O.relu != 0;
        oacc = O.accumulate != 0;
        ogsum = O.gate_sum != 0;
      }
    };
    auto step_out = [&]() {
      och += 16;
      if (och == oclen) {
        och = 0;
        ++ov;
      }
    };
    f32x4 X[QC];
    for (int pass = 0; pass < a.n_pass; ++pass) {
      f32x4 acc[NCBP];
#pragma unroll
      for (int cb = 0; cb < NCBP; ++cb) acc[cb] = *reinterpret_cast<const f32x4*>(&b_lds[16 * (pass * NCBP + cb) + 4 * g]);
      int iv = 0, ich = 0, iclen = 0;  // input: view and channel of the next 16-channel group
      const float* in_ptr = nullptr;
      for (int chunk = 0; chunk < a.n_kchunk; ++chunk) {
        if (pass == 0 || a.n_kchunk > 1) {  // (a single chunk stays in its registers for every pass)
#pragma unroll
          for (int q = 0; q < QC; ++q) {
            if (ich == 0) {  // uniform
              in_ptr = d.in[iv].ptr + view_pixel_offset32(d.in[iv], n, y, x) + 4 * g;
              iclen = d.in[iv].c_len;
            }
            X[q] = *reinterpret_cast<const f32x4*>(in_ptr + ich);
            ich += 16;
            if (ich == iclen) {
              ich = 0;
              ++iv;
            }
          }
        }
        // fragment reads by hand, one step ahead (asm volatile: the reads are invariant across tiles and hipcc would
        // hoist all of them out of the tile loop -- 4 QC NCBP registers -- and spill)
        const unsigned wb = wa + static_cast<unsigned>((pass * a.n_kchunk + chunk) * (QC * NCBP * 1024));
        f32x4 wf[2];
        asm volatile("ds_read_b128 %0, %1" : "=v"(wf[0]) : "v"(wb) : "memory");
        static_for<QC * NCBP>([&](auto ic) {
          constexpr int i = decltype(ic)::v, q = i / NCBP, cb = i % NCBP;
          if constexpr (i + 1 < QC * NCBP) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[(i + 1) & 1]) : "v"(wb), "n"((i + 1) * 1024) : "memory");
            asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wf[i & 1]));
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[i & 1]));
          }
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[cb] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[i & 1][e], X[q][e], acc[cb], 0, 0, 0);
        });
      }
      // ---- epilogue of the pass: register e of acc[cb] = column 16 (pass NCBP + cb) + 4 g + e of this lane's pixel ----
      if constexpr (EPI == 0) {
#pragma unroll
        for (int cb = 0; cb < NCBP; ++cb) {
          if (och == 0) enter_out();  // uniform
          if (a.nt_store) __builtin_nontemporal_store(acc[cb], reinterpret_cast<f32x4*>(optr + och));
          else *reinterpret_cast<f32x4*>(optr + och) = acc[cb];
          step_out();
        }
      } else {
        // groups of up to four blocks: the group's gate / previous-value reads first, then its stores (vmcnt counts in
        // order: a read issued between two stores would wait for the store before it)
        constexpr int GRP = NCBP < 4 ? NCBP : 4;
#pragma unroll
        for (int c0 = 0; c0 < NCBP; c0 += GRP) {
          float* dst[GRP];
          f32x4 gt[GRP], old[GRP];
          bool has_gate[GRP], relu[GRP], accum[GRP], gsum[GRP];
#pragma unroll
          for (int j = 0; j < GRP; ++j) {
            if (och == 0) enter_out();
            dst[j] = optr + och;
            has_gate[j] = ogate != nullptr;
            relu[j] = orelu;
            accum[j] = oacc;
            gsum[j] = ogsum;
            if (has_gate[j]) gt[j] = *reinterpret_cast<const f32x4*>(ogate + och);  // uniform branches
            if (accum[j]) old[j] = *reinterpret_cast<const f32x4*>(dst[j]);
            step_out();
          }
#pragma unroll
          for (int j = 0; j < GRP; ++j) {
            f32x4 v = acc[c0 + j];
            if (relu[j]) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (has_gate[j] && !gsum[j]) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (gt[j][e] > 0.f) ? v[e] : 0.f;
            }
            if (accum[j]) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] += old[j][e];
            }
            if (has_gate[j] && gsum[j]) {
#pragma unroll
              for (int e = 0; e < 4; ++e) v[e] = (gt[j][e] > 0.f) ? v[e] : 0.f;
            }
            if (a.nt_store) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst[j]));
            else *reinterpret_cast<f32x4*>(dst[j]) = v;
          }
        }
      }
    }
  }
}

bool pw_in_view_ok(const unetpp_view& v) {
  return v.scale == nullptr && v.shift == nullptr && v.gate == nullptr && v.relu == 0 && (v.c_len & 15) == 0 &&
         ((v.C | v.c_off) & 3) == 0 && (reinterpret_cast<uintptr_t>(v.ptr) & 15) == 0;
}

bool pw_out_view_ok(const unetpp_view& v, bool several) {
  return (v.c_len & (several ? 31 : 15)) == 0 && ((v.C | v.c_off) & 3) == 0 && (reinterpret_cast<uintptr_t>(v.ptr) & 15) == 0 &&
         (v.gate == nullptr || (reinterpret_cast<uintptr_t>(v.gate) & 15) == 0);
}

int block_count(int units16) {  // 16-wide groups per chunk / blocks per pass: all of them up to 8, else 8 with a whole number of rounds
  if (units16 == 1 || units16 == 2 || units16 == 4 || units16 == 8) return units16;
  return (units16 > 8 && (units16 & 7) == 0) ? 8 : 0;
}

template <int QC, int NCBP, int EPI>
int launch_pw_epi(const PwArgs& a, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st) {
  static bool raised[64] = {};  // per device: dynamic LDS above 64 KB has to be allowed once per function
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return UNETPP_ELAUNCH;
  if (!raised[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pw_kernel<QC, NCBP, EPI>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            160 * 1024) != hipSuccess)
      return UNETPP_ELAUNCH;
    raised[dev] = true;
  }
  hipLaunchKernelGGL((gemm_pw_kernel<QC, NCBP, EPI>), grid, block, lds_bytes, st, a);
  return launch_status();
}

template <int QC, int NCBP>
int launch_pw(const PwArgs& a, bool plain, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st) {
  return plain ? launch_pw_epi<QC, NCBP, 0>(a, grid, block, lds_bytes, st) : launch_pw_epi<QC, NCBP, 1>(a, grid, block, lds_bytes, st);
}

}  // namespace

// returns UNETPP_OK after launching, 1 when the descriptor is not one this kernel takes (gemm_fast.hip then does)
int launch_gemm_pw(const unetpp_gemm_desc* d, const FastArgs& fa, hipStream_t st) {
  if (d->taps != 1 || (d->flags & UNETPP_GEMM_BF16) != 0 || d->weight_image == nullptr || d->stats_partial != nullptr) return 1;
  if (opt_value(OPT_PW_DIRECT, 1) == 0 || (d->W & 15) != 0) return 1;
  PwArgs a;
  a.d = *d;
  a.K = fa.Ktot;
  a.N = fa.Ncols;
  for (int i = 0; i < d->n_in; ++i)
    if (!pw_in_view_ok(d->in[i])) return 1;
  for (int i = 0; i < d->n_out; ++i)
    if (!pw_out_view_ok(d->out[i], d->n_out > 1)) return 1;
  const int qc = block_count(a.K >> 4), ncbp = block_count(a.N >> 4);
  if (qc == 0 || ncbp == 0) return 1;
  const size_t lds_bytes = (static_cast<size_t>(a.K) * a.N + a.N) * sizeof(float);
  // (Weights beyond the LDS -- level 2 of the base-32 network: 512 KB -- were tried as column groups, one 128 KB group per
  // workgroup of sixteen waves: 139 / 147 us against gemm_fast_kernel<1>'s 112 / 122, profiles/r6/bench_pw_column_groups.txt:
  // two tiles per wave do not pay for staging the group.)
  if (lds_bytes > 148 * 1024) return 1;
  a.n_kchunk = (a.K >> 4) / qc;
  a.n_pass = (a.N >> 4) / ncbp;
  a.tiles_x = d->W >> 4;
  a.tiles_shift = -1;
  for (int s = 0; s < 16; ++s)
    if ((1 << s) == a.tiles_x) a.tiles_shift = s;
  a.n_tiles = static_cast<long>(d->N) * d->H * a.tiles_x;
  if (a.n_tiles >= 0x7fffffffL) return 1;
  // outputs far beyond the L2 (4 MB per XCD) leave as non-temporal stores; OPT_PW_NT forces either form
  long out_bytes = 0;
  for (int i = 0; i < d->n_out; ++i) out_bytes += static_cast<long>(d->N) * d->H * d->W * d->out[i].c_len * 4;
  a.nt_store = static_cast<int>(opt_value(OPT_PW_NT, out_bytes >= (64L << 20) ? 1 : 0));
  const int cus = device_cu_count();
  if (cus <= 0) return UNETPP_ELAUNCH;
  // 16 waves per CU: four workgroups of four waves while four weight images fit the LDS, else two of eight, else one of 16
  const int threads = lds_bytes <= 38 * 1024 ? 256 : (lds_bytes <= 78 * 1024 ? 512 : 1024);
  const int waves = threads >> 6;
  long blocks = static_cast<long>(cus) * (16 / waves);
  const long need = (a.n_tiles + waves - 1) / waves;
  if (blocks > need) blocks = need;
  const dim3 grid(static_cast<unsigned>(blocks)), block(threads);
  bool plain = true;  // no output view asks for more than bias + store
  for (int i = 0; i < d->n_out; ++i)
    plain = plain && d->out[i].relu == 0 && d->out[i].accumulate == 0 && d->out[i].gate == nullptr;
  int rc = 1;
#define UNETPP_PW_CASE(Q, C) \
  if (qc == Q && ncbp == C) rc = launch_pw<Q, C>(a, plain, grid, block, lds_bytes, st);
#define UNETPP_PW_ROW(Q) UNETPP_PW_CASE(Q, 1) UNETPP_PW_CASE(Q, 2) UNETPP_PW_CASE(Q, 4) UNETPP_PW_CASE(Q, 8)
  UNETPP_PW_ROW(1) UNETPP_PW_ROW(2) UNETPP_PW_ROW(4) UNETPP_PW_ROW(8)
#undef UNETPP_PW_ROW
#undef UNETPP_PW_CASE
  if (rc == UNETPP_OK) note_kernel("gemm_pw_kernel");
  return rc;
}

}  // namespace unetpp

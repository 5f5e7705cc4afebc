// bf16-storage twin of gemm_pw.hip: pointwise (taps = 1) GEMM of the transposed convolutions / 1x1 convolutions of
// BASELINE configs[3] / configs[4] with the weights resident in LDS and the activations loaded straight into the MFMA
// operand registers (/root/reference/models/unet.py:187,191).  At these widths the launches are byte movers (16 MFMA
// instructions of 16 cycles per 6 KB of a 16-pixel tile at K 64, N 128): what counts is that every byte is requested
// once, in 64-byte runs, by sixteen waves per CU that never meet at a barrier.
//   * v_mfma_f32_16x16x32_bf16: A = weights (rows = output columns), B = activations (columns = pixels); lane
//     (pixel t16, k-group kg) loads 16 bytes = channels 32 q + 8 kg .. + 7 of its pixel;
//   * two 16-row blocks share a 32-column group so that the eight accumulator registers of a lane are EIGHT CONSECUTIVE
//     output channels (block 0: columns 8 kg + rr, block 1: columns 8 kg + 4 + rr of the group): one 16-byte bf16 store
//     per pixel and group straight from the registers; bias, ReLU, gate and accumulation in fp32 before the rounding;
//   * the weights come from the image gemm_bf16.hip / gemm_bf16_dma.hip use (weight_image.hip, kind "bf16":
//     [column tile 32][chunk 32][g 2][col 32][h 2][8 bf16]) and are re-ordered while they are staged.
#include "bf16_common.h"
#include "common.h"
#include "gemm_units.h"
#include "lds_asm.h"

namespace unetpp {
namespace {

struct PwBfArgs {
  unetpp_gemm_desc d;
  int K, N;
  int tiles_x, tiles_shift;
  int n_pass, n_kchunk;
  long n_tiles;
};

// QC = 32-channel groups per K chunk (4 QC fragment registers), NCBP = 32-column groups per column pass (8 NCBP
// accumulator registers); EPI 0: plain stores, 1: ReLU / gate / accumulate per output view
template <int QC, int NCBP, int EPI>
__global__ __launch_bounds__(1024, 1) void gemm_pw_bf16_kernel(const PwBfArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pwb_lds[];
  unsigned char* w_lds = pwb_lds;
  float* b_lds = reinterpret_cast<float*>(pwb_lds + static_cast<size_t>(a.K) * a.N * 2);
  const unetpp_gemm_desc& d = a.d;
  const int tid = threadIdx.x, lane = tid & 63, t16 = lane & 15, kg = lane >> 4;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), n_waves = blockDim.x >> 6;

  // ---- weights -> LDS, once: block (pass, chunk) holds [q QC][P NCBP][blk 2][kg 4][t16 16] x 16 bytes =
  // W[k = 32 (chunk QC + q) + 8 kg .. + 7][n = 32 (pass NCBP + P) + 8 (t16 >> 2) + 4 blk + (t16 & 3)] ----
  {
    const int NP = a.N >> 5, NQ = a.K >> 5;
    const u32x4* img = reinterpret_cast<const u32x4*>(d.weight_image);
    for (int i = tid; i < NQ * NP * 128; i += blockDim.x) {
      const int l = i & 63, blk = (i >> 6) & 1, pq = i >> 7;
      const int pg = pq % NP, qg = pq / NP;
      const int k4 = l >> 4, tt = l & 15;
      const int col = 8 * (tt >> 2) + 4 * blk + (tt & 3), gg = k4 >> 1, h = k4 & 1;
      const u32x4 v = img[(static_cast<long>(pg) * NQ + qg) * 128 + (gg * 32 + col) * 2 + h];
      const int pass = pg / NCBP, P = pg - pass * NCBP, chunk = qg / QC, q = qg - chunk * QC;
      reinterpret_cast<u32x4*>(w_lds)[((((pass * a.n_kchunk + chunk) * QC + q) * NCBP + P) * 2 + blk) * 64 + l] = v;
    }
    for (int i = tid; i < a.N; i += blockDim.x) b_lds[i] = d.bias != nullptr ? d.bias[i] : 0.f;
  }
  __syncthreads();

  const unsigned wa = static_cast<unsigned>(reinterpret_cast<uintptr_t>(w_lds)) + lane * 16;
  const long stride = static_cast<long>(gridDim.x) * n_waves;
  for (long tl = static_cast<long>(blockIdx.x) * n_waves + wave; tl < a.n_tiles; tl += stride) {
    const unsigned t = static_cast<unsigned>(tl);
    const unsigned row = a.tiles_shift >= 0 ? t >> a.tiles_shift : t / static_cast<unsigned>(a.tiles_x);
    const int x = static_cast<int>(t - row * a.tiles_x) * 16 + t16;
    const int n = static_cast<int>(row / static_cast<unsigned>(d.H)), y = static_cast<int>(row - n * d.H);

    int ov = 0, och = 0;  // output cursor: view and channel of the next 32-column group
    bf16_t* optr = nullptr;
    const bf16_t* ogate = nullptr;
    int oclen = 0;
    bool orelu = false, oacc = false, ogsum = false;
    auto enter_out = [&]() {
      const unetpp_view& O = d.out[ov];
      const unsigned off = view_pixel_offset32(O, n, y, x) + 8 * kg;
      optr = reinterpret_cast<bf16_t*>(O.ptr) + off;
      oclen = O.c_len;
      if constexpr (EPI != 0) {
        ogate = O.gate != nullptr ? reinterpret_cast<const bf16_t*>(O.gate) + off : nullptr;
        orelu = O.relu != 0;
        oacc = O.accumulate != 0;
        ogsum = O.gate_sum != 0;
      }
    };
    auto step_out = [&]() {
      och += 32;
      if (och == oclen) {
        och = 0;
        ++ov;
      }
    };
    u32x4 X[QC];
    for (int pass = 0; pass < a.n_pass; ++pass) {
      f32x4 acc[NCBP][2];
#pragma unroll
      for (int P = 0; P < NCBP; ++P)
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
          acc[P][blk] = *reinterpret_cast<const f32x4*>(&b_lds[32 * (pass * NCBP + P) + 8 * kg + 4 * blk]);
      int iv = 0, ich = 0, iclen = 0;
      const bf16_t* in_ptr = nullptr;
      for (int chunk = 0; chunk < a.n_kchunk; ++chunk) {
        if (pass == 0 || a.n_kchunk > 1) {  // (a single chunk stays in its registers for every pass)
#pragma unroll
          for (int q = 0; q < QC; ++q) {
            if (ich == 0) {  // uniform
              in_ptr = reinterpret_cast<const bf16_t*>(d.in[iv].ptr) + view_pixel_offset32(d.in[iv], n, y, x) + 8 * kg;
              iclen = d.in[iv].c_len;
            }
            X[q] = *reinterpret_cast<const u32x4*>(in_ptr + ich);
            ich += 32;
            if (ich == iclen) {
              ich = 0;
              ++iv;
            }
          }
        }
        // fragment reads by hand, one step ahead (asm volatile: invariant across tiles, hipcc would hoist and spill them)
        const unsigned wb = wa + static_cast<unsigned>((pass * a.n_kchunk + chunk) * (QC * NCBP * 2048));
        const unsigned wb_hi = wb + 65536;  // (a ds_read offset has 16 bits: fragments 64 .. 127 of an 8 x 8 block)
        u32x4 wf[2];
        asm volatile("ds_read_b128 %0, %1" : "=v"(wf[0]) : "v"(wb) : "memory");
        static_for<QC * NCBP * 2>([&](auto ic) {
          constexpr int i = decltype(ic)::v, q = i / (NCBP * 2), P = (i / 2) % NCBP, blk = i & 1;
          if constexpr (i + 1 < QC * NCBP * 2) {
            asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(wf[(i + 1) & 1]) : "v"(i + 1 < 64 ? wb : wb_hi), "n"(((i + 1) & 63) * 1024) : "memory");
            asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(wf[i & 1]));
          } else {
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(wf[i & 1]));
          }
          acc[P][blk] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, wf[i & 1]), __builtin_bit_cast(bf16x8, X[q]),
                                                                acc[P][blk], 0, 0, 0);
        });
      }
      // ---- epilogue of the pass: acc[P][blk][rr] = column 32 (pass NCBP + P) + 8 kg + 4 blk + rr of this lane's pixel ----
      if constexpr (EPI == 0) {
#pragma unroll
        for (int P = 0; P < NCBP; ++P) {
          if (och == 0) enter_out();  // uniform
          const float v[8] = {acc[P][0][0], acc[P][0][1], acc[P][0][2], acc[P][0][3], acc[P][1][0], acc[P][1][1], acc[P][1][2], acc[P][1][3]};
          *reinterpret_cast<u32x4*>(optr + och) = pack8(v);
          step_out();
        }
      } else {
        constexpr int GRP = NCBP < 2 ? NCBP : 2;  // (two groups of reads in flight: four spill beside 8 x 4 blocks)
#pragma unroll
        for (int c0 = 0; c0 < NCBP; c0 += GRP) {
          bf16_t* dst[GRP];
          u32x4 gt[GRP], old[GRP];
          bool has_gate[GRP], relu[GRP], accum[GRP], gsum[GRP];
#pragma unroll
          for (int j = 0; j < GRP; ++j) {
            if (och == 0) enter_out();
            dst[j] = optr + och;
            has_gate[j] = ogate != nullptr;
            relu[j] = orelu;
            accum[j] = oacc;
            gsum[j] = ogsum;
            if (has_gate[j]) gt[j] = *reinterpret_cast<const u32x4*>(ogate + och);  // uniform branches
            if (accum[j]) old[j] = *reinterpret_cast<const u32x4*>(dst[j]);
            step_out();
          }
#pragma unroll
          for (int j = 0; j < GRP; ++j) {
            float v[8] = {acc[c0 + j][0][0], acc[c0 + j][0][1], acc[c0 + j][0][2], acc[c0 + j][0][3],
                          acc[c0 + j][1][0], acc[c0 + j][1][1], acc[c0 + j][1][2], acc[c0 + j][1][3]};
            float gv[8], ov8[8];
            if (has_gate[j]) unpack8(gt[j], gv);
            if (accum[j]) unpack8(old[j], ov8);
            if (relu[j]) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
            }
            if (has_gate[j] && !gsum[j]) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (gv[e] > 0.f) ? v[e] : 0.f;
            }
            if (accum[j]) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += ov8[e];
            }
            if (has_gate[j] && gsum[j]) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (gv[e] > 0.f) ? v[e] : 0.f;
            }
            *reinterpret_cast<u32x4*>(dst[j]) = pack8(v);
          }
        }
      }
    }
  }
}

bool pwb_in_ok(const unetpp_view& v) {
  return v.scale == nullptr && v.shift == nullptr && v.gate == nullptr && v.relu == 0 && (v.c_len & 31) == 0 && bf16_view_aligned(v);
}
bool pwb_out_ok(const unetpp_view& v) { return (v.c_len & 31) == 0 && bf16_view_aligned(v); }

int pwb_block_count(int units) {
  if (units == 1 || units == 2 || units == 4 || units == 8) return units;
  return (units > 8 && (units & 7) == 0) ? 8 : 0;
}

template <int QC, int NCBP, int EPI>
int launch_pwb_epi(const PwBfArgs& a, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st) {
  static bool raised[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return UNETPP_ELAUNCH;
  if (!raised[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pw_bf16_kernel<QC, NCBP, EPI>),
                            hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess)
      return UNETPP_ELAUNCH;
    raised[dev] = true;
  }
  hipLaunchKernelGGL((gemm_pw_bf16_kernel<QC, NCBP, EPI>), grid, block, lds_bytes, st, a);
  return launch_status();
}
template <int QC, int NCBP>
int launch_pwb(const PwBfArgs& a, bool plain, dim3 grid, dim3 block, size_t lds_bytes, hipStream_t st) {
  if (plain) return launch_pwb_epi<QC, NCBP, 0>(a, grid, block, lds_bytes, st);
  if constexpr (NCBP <= 4) return launch_pwb_epi<QC, NCBP, 1>(a, grid, block, lds_bytes, st);
  return 1;
}

}  // namespace

// returns UNETPP_OK after launching, 1 when the descriptor is not one this kernel takes (fa = bf16_gemm_args of d)
int launch_gemm_pw_bf16(const unetpp_gemm_desc* d, const FastArgs& fa, hipStream_t st) {
  if (d->taps != 1 || (d->flags & UNETPP_GEMM_BF16) == 0 || d->weight_image == nullptr || d->stats_partial != nullptr) return 1;
  if (opt_value(OPT_PW_DIRECT, 1) == 0 || (d->W & 15) != 0) return 1;
  PwBfArgs a;
  a.d = *d;
  a.K = fa.Ktot;
  a.N = fa.Ncols;
  for (int i = 0; i < d->n_in; ++i)
    if (!pwb_in_ok(d->in[i])) return 1;
  for (int i = 0; i < d->n_out; ++i)
    if (!pwb_out_ok(d->out[i])) return 1;
  bool plain = true;
  for (int i = 0; i < d->n_out; ++i)
    plain = plain && d->out[i].relu == 0 && d->out[i].accumulate == 0 && d->out[i].gate == nullptr;
  int qc = pwb_block_count(a.K >> 5), ncbp = pwb_block_count(a.N >> 5);
  if (qc == 0 || ncbp == 0) return 1;
  if (!plain && ncbp == 8) ncbp = 4;  // (the read-modify-write epilogue beside 64 accumulators does not fit 128 registers)
  const size_t lds_bytes = static_cast<size_t>(a.K) * a.N * 2 + static_cast<size_t>(a.N) * 4;
  if (lds_bytes > 148 * 1024) return 1;
  a.n_kchunk = (a.K >> 5) / qc;
  a.n_pass = (a.N >> 5) / ncbp;
  a.tiles_x = d->W >> 4;
  a.tiles_shift = -1;
  for (int s = 0; s < 16; ++s)
    if ((1 << s) == a.tiles_x) a.tiles_shift = s;
  a.n_tiles = static_cast<long>(d->N) * d->H * a.tiles_x;
  if (a.n_tiles >= 0x7fffffffL) return 1;
  const int cus = device_cu_count();
  if (cus <= 0) return UNETPP_ELAUNCH;
  const int threads = lds_bytes <= 38 * 1024 ? 256 : (lds_bytes <= 78 * 1024 ? 512 : 1024);
  const int waves = threads >> 6;
  long blocks = static_cast<long>(cus) * (16 / waves);
  const long need = (a.n_tiles + waves - 1) / waves;
  if (blocks > need) blocks = need;
  const dim3 grid(static_cast<unsigned>(blocks)), block(threads);
  int rc = 1;
#define UNETPP_PWB_CASE(Q, C) \
  if (qc == Q && ncbp == C) rc = launch_pwb<Q, C>(a, plain, grid, block, lds_bytes, st);
#define UNETPP_PWB_ROW(Q) UNETPP_PWB_CASE(Q, 1) UNETPP_PWB_CASE(Q, 2) UNETPP_PWB_CASE(Q, 4) UNETPP_PWB_CASE(Q, 8)
  UNETPP_PWB_ROW(1) UNETPP_PWB_ROW(2) UNETPP_PWB_ROW(4) UNETPP_PWB_ROW(8)
#undef UNETPP_PWB_ROW
#undef UNETPP_PWB_CASE
  if (rc == UNETPP_OK) note_kernel("gemm_pw_bf16_kernel");
  return rc;
}

}  // namespace unetpp

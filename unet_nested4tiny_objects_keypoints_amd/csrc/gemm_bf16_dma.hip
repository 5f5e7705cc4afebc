// bf16-storage multi-view pixel GEMM with BOTH operands staged by LDS-DMA (round 3; BASELINE configs[3]/[4]).
//
// gemm_bf16.hip stages through registers: per 32-channel chunk a thread issues 11 global loads, unpacks / masks /
// re-packs them and writes 11 LDS stores -- ~600 vector and ~340 scalar instructions per unit and wave around 36 MFMAs of
// 32 cycles (SQ counters, profiles/r3/pmc_sq_bf16_*: matrix pipe 0.25 busy, the level-0 layers stream 2.5-4.4 TB/s).
// At bf16 MFMA speed every layer with fewer than ~300 FLOP per byte is a streaming problem, so this kernel removes the
// instruction stream instead of tuning it:
//   * the input patch goes HBM/L2 -> LDS by `buffer_load_dwordx4 ... lds` (1 KB per wave instruction, no staging
//     registers, no VALU): the buffer resource's range check returns ZEROS for out-of-image halo pixels (their offset is
//     simply out of range), so border patches take the same path as interior ones -- no clamps, masks or selects;
//   * an LDS-DMA instruction writes its 64 lanes' 16 bytes lane-linearly, so the padded 80-byte pixel pitch of
//     gemm_bf16.hip is not available; instead the SOURCE side is permuted: a 1 KB block holds 16 consecutive halo
//     pixels x 4 channel octets as [octet][pixel], i.e. lane L fetches pixel 16*block + (L & 15), octet L >> 4.  The 16
//     lanes of a ds_read_b128 phase then read 16 consecutive pixels of one octet = 256 contiguous bytes: conflict free;
//   * all input views of a launch share one geometry (C, Hs, Ws, strides: the dense-skip concatenation of tensors of one
//     level, the four phases of a transposed convolution), so a thread's offsets depend on the pixel patch only; view,
//     phase origin and channel chunk go into the scalar offset / the resource;
//   * the weight image of the chunk follows by `global_load_lds`; both operands are double buffered in LDS, the DMA of
//     chunk c+1 is issued before the MFMAs of chunk c, one barrier per chunk, two workgroups per CU (2 x 80 KB);
//   * epilogues as in gemm_bf16.hip (swapped MFMA operands + v_permlane32_swap for plain launches, pixel-major
//     accumulators + LDS transpose for the BatchNorm-statistics launches).
// Taken for: plain input views (no BatchNorm fold / ReLU / gate on load -- those need VALU on the way and stay with
// gemm_bf16.hip), channel slices in multiples of 32, tensors below 2 GB.  Same weight image, same results bit for bit
// (tests/test_gpu_bf16.py::test_bf16_dma_and_register_kernels_agree).
//
// What it showed (tools/bf16_dma_ablation.sh; corrected in round 5: the NO_EPI variant of rounds 3-4 dropped the epilogue
// call and nothing else, so hipcc deleted every MFMA with it -- its rows were "no epilogue AND no matrix phase".  With
// the accumulators kept live, profiles/r5/ablation_gemm_bf16_dma_stores_c3.txt / _c5.txt): 32 -> 32 at 512 x 512 x 8
// (HBM floor 34 us) 73 us as it stands; without the output stores 61, without the whole epilogue 50, without the MFMAs
// 67, without the input DMA 57.  [32 x 4] -> 32: 196 us; 187 / 170 / 156 / 128.  [64 x 5] -> 64 at 384 x 384 x 4
// (configs[4]): 231 us; 222 / 217 / 164 / 156, without the weight DMA 199.  So: one-chunk launches lose a third to the
// epilogue (half of it the stores themselves), long-K launches 6 %; there the input DMA and the matrix phase are the two
// large terms and they overlap only partly -- a chunk's 75 KB arrive at ~16 GB/s per CU (4.2 TB/s over the chip, L2
// hits included), which is the rate a plain streaming kernel gets from HBM, while its MFMAs need 1.9 us of the 4.6.
// Neither contiguous 1 KB wave stores nor issuing the next DMA and the next unit's epilogue operands BEFORE a unit's
// stores (counted vmcnt, two operand sets; 228 registers) changed the sum; what is left is bytes per pixel (halo,
// weights per chunk), not issue order.
//
// Round 4, measured on ONE box with alternating libraries (tools/dma_ab.sh; boxes of the pool differ by up to 30 % on these
// store-heavy launches, which is how the first of these looked like a 3-7 % gain when it was timed on another box):
//   * a quad-coalesced patch image (a lane quad fetches one pixel's 64 contiguous bytes, XOR-swizzled so that the
//     fragment reads stay conflict free) instead of the [octet][pixel] blocks: +-1 %, 11 more registers.  Not kept;
//   * the ten DMA pieces of a chunk issued one behind each MFMA step instead of as a block in front of them: -1 % on the
//     forward GEMMs of configs[4], 0 / +2 % elsewhere, 256 registers + scratch.  Not kept;
//   * a second chunk of DMA in flight (experiment build): no change; non-temporal output stores: 5-30 % slower.
//   * deferred output stores for the forward launches (no gate / accumulate operands: their 64 registers hold the packed
//     tile, whose eight store instructions are issued one behind every second MFMA step of the NEXT chunk): 1.4-1.9 %
//     SLOWER on both bf16 steps -- once more: what a CU's memory path moves per unit is the bound, not when it moves it.
// tools/dma_stamps.py (stamped build): per 32-channel chunk of [64 x 5] -> 64 at 384 x 384 x 4 wave 0 spends 600 cycles in
// the cursor, 1 860 issuing its ten DMA pieces, 2 650 in LDS reads + MFMAs, 435 in the epilogue (per-chunk average) and
// 2 450 in the barrier, most of which is the second wave of its SIMD running ITS MFMAs.
#include <cstdlib>

#include "bf16_common.h"
#include "common.h"
#include "gemm_units.h"
#include "lds_asm.h"
#include "dma_experiments.h"

namespace unetpp {
namespace {

// In-kernel phase stamps (profiling builds only: -DUNETPP_DMA_STAMPS, tools/dma_stamps.py): wave 0 of every workgroup adds
// the cycles it spent in each phase of its (unit, chunk) stream to a global table.
#ifdef UNETPP_DMA_STAMPS
__device__ unsigned long long g_dma_stamps[16];
#define DMA_STAMP(i)                                            \
  do {                                                          \
    const unsigned long long now_ = __builtin_amdgcn_s_memtime(); \
    st_acc[i] += now_ - st_last;                                \
    st_last = now_;                                             \
  } while (0)
#else
#define DMA_STAMP(i) \
  do {               \
  } while (0)
#endif

constexpr int DKC = 32;        // channels per K chunk (64 bytes per pixel = 4 octets of 16 bytes)
constexpr int DSTEP = 1024;    // bytes of one (tap, g) weight step: 32 columns x 16 k x 2 B
constexpr unsigned kDmaOutOfRange = 0x80000000u;  // buffer offset no view reaches (tensors are below 2 GB)
typedef const __attribute__((address_space(1))) void* dma_gptr_t;
typedef __attribute__((address_space(3))) void* dma_lptr_t;

struct DmaArgs {
  FastArgs f;
  int pitch;        // bytes per tensor pixel of the input views (C * 2)
  int row_pitch;    // bytes per tensor row step of one logical row (sy * Ws * C * 2)
  int col_pitch;    // bytes per logical column step (sx * C * 2)
  int img_pitch;    // bytes per image / 1 (Hs * Ws * C * 2), < 2^31
  int view_bytes;   // size of an input tensor in bytes
  int wimg_bytes;   // size of the launch's weight image in bytes (n_tiles * n_chunks * IMG)
  int unit_base;    // first unit of this launch in the enumeration of its form (split launches: launch_gemm_bf16_dma)
};

// WAVES = 4: 256-pixel patches (8 x 32), two workgroups per CU (2 x 80 KB).  WAVES = 8: 512-pixel patches (16 x 32:
// 1.19x halo instead of 1.33x), one workgroup per CU, up to two column tiles per staged patch and a 72 KB weight region
// that holds the launch's WHOLE weight image when it has at most four (tile, chunk) images (K = 128 -> 32, 64 -> 64):
// everything that passes through a CU's vector memory path costs the same whether it comes from HBM or hits the L2, so
// the bytes per pixel -- halo, patch re-staging per column group, weight re-streaming per chunk -- are what to cut.
// RMW = false (round 5, pointwise launches only): no output view has a ReLU gate or an accumulate flag -- the forward of a
// transposed convolution / 1x1 convolution.  The epilogue then carries no gate / previous-value operands (64 registers for
// two column tiles), the kernel fits 168 registers and THREE workgroups share a CU: a pointwise unit is 16 MFMAs per wave
// behind 40 KB of DMA, so what it needs is more bytes in flight per CU, and a third workgroup is the cheapest way to get
// them (its LDS is 40 KB).
template <int TAPS, int LOG2TW, int NT, bool STATS, int WAVES = 4, bool RMW = true>
__global__ __launch_bounds__(64 * WAVES, RMW ? 2 : 3) void gemm_bf16_dma_kernel(const DmaArgs da) {
  const FastArgs& a = da.f;
  constexpr int THREADS = 64 * WAVES, PIX = 64 * WAVES;
  static_assert(WAVES == 4 || (WAVES == 8 && TAPS == 9 && !STATS), "the 8-wave form: plain 3x3 launches");
  static_assert(RMW || (TAPS == 1 && WAVES == 4 && !STATS), "the three-per-CU form: plain pointwise launches");
  constexpr int HALO = (TAPS == 9) ? 1 : 0;
  constexpr int TW = 1 << LOG2TW, TH = PIX >> LOG2TW;
  constexpr int HWp = TW + 2 * HALO, HHp = TH + 2 * HALO;
  constexpr int NPIX = HWp * HHp;
  constexpr int NBLK = (NPIX + 15) / 16;        // 1 KB blocks of 16 pixels x 4 octets
  constexpr int IN_BYTES = NBLK * 1024;         // 22528 (3x3) / 16384 (pointwise): >= the 4 x 4 KB epilogue scratch + 1 KB
  constexpr int IMG = TAPS * 2 * DSTEP;         // bytes of one (column tile, chunk) image
  constexpr int W_BYTES = NT * IMG;             // the NT images of one chunk
  constexpr int W_SLOTS = (WAVES == 8) ? 4 / NT : 2;   // chunks the weight region holds (two of them = the streaming ring)
  constexpr int NQ = (NBLK + WAVES - 1) / WAVES;       // input blocks per wave
  constexpr int WBLK = W_BYTES / 1024, NWQ = (WBLK + WAVES - 1) / WAVES;
  static_assert(!STATS || IN_BYTES >= 4 * 4096 + 1024, "epilogue scratch + statistics rows do not fit in an input buffer");
  static_assert(NBLK >= WAVES, "input staging: a wave past the patch repeats block `wave`, which must exist");
  static_assert(2 * IN_BYTES + W_SLOTS * W_BYTES <= (WAVES == 4 ? 80 : 160) * 1024, "LDS: two workgroups (one) per CU");
  __shared__ __attribute__((aligned(1024))) unsigned char smem[2 * IN_BYTES + W_SLOTS * W_BYTES];

  const unetpp_gemm_desc& d = a.d;
  const int tid = threadIdx.x;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63, j = lane & 31, h = lane >> 5;

  constexpr bool CONTIG = TAPS == 9;  // (pointwise units are short and write strided phase views: round robin inside an XCD)
  UnitRange ur = CONTIG ? my_contiguous_unit_range(a.total_blocks) : my_unit_range(a.total_blocks);
  ur.first += da.unit_base;
  const int my_units = __builtin_amdgcn_readfirstlane(static_cast<int>(ur.count));
  if (my_units == 0) return;
  long p_index = ur.first, c_index = ur.first;
  auto decode = [&](long lb) {  // (gemm_units.h's decode_unit with this kernel's patch height; a.tiles_y counts TH-row patches)
    UnitGeom g;
    unsigned bid = static_cast<unsigned>(lb);
    g.group = static_cast<int>(bid % static_cast<unsigned>(a.n_groups));
    bid /= static_cast<unsigned>(a.n_groups);
    g.patch = bid;
    const int txi = static_cast<int>(bid % static_cast<unsigned>(a.tiles_x));
    bid /= static_cast<unsigned>(a.tiles_x);
    const int tyi = static_cast<int>(bid % static_cast<unsigned>(a.tiles_y));
    g.n = static_cast<int>(bid / static_cast<unsigned>(a.tiles_y));
    g.ty0 = tyi * TH;
    g.tx0 = txi * TW;
    return g;
  };
  auto step_unit = [&](UnitGeom& u, long& index) {
    if constexpr (!CONTIG) {
      index += ur.step;
      u = decode(index);
      u.n = __builtin_amdgcn_readfirstlane(u.n);
      u.ty0 = __builtin_amdgcn_readfirstlane(u.ty0);
      u.tx0 = __builtin_amdgcn_readfirstlane(u.tx0);
      u.group = __builtin_amdgcn_readfirstlane(u.group);
      u.patch = __builtin_amdgcn_readfirstlane(static_cast<int>(u.patch));
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
  UnitGeom p_ug = decode(ur.first);
  p_ug.n = __builtin_amdgcn_readfirstlane(p_ug.n);
  p_ug.ty0 = __builtin_amdgcn_readfirstlane(p_ug.ty0);
  p_ug.tx0 = __builtin_amdgcn_readfirstlane(p_ug.tx0);
  p_ug.group = __builtin_amdgcn_readfirstlane(p_ug.group);
  p_ug.patch = __builtin_amdgcn_readfirstlane(static_cast<int>(p_ug.patch));  // < 2^31 (fast_args)
  UnitGeom c_ug = p_ug;

  // ---- compute side: LDS byte offsets of this lane's A fragments, per pixel tile and tap (octet h of the chunk half g) ----
  int a_off[2][TAPS];
#pragma unroll
  for (int mt = 0; mt < 2; ++mt) {
    const int p = 64 * wave + 32 * mt + j;
    const int py = p >> LOG2TW, px = p & (TW - 1);
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
      const int hp = (py + (TAPS == 9 ? tap / 3 : 0)) * HWp + px + (TAPS == 9 ? tap % 3 : 0);
      a_off[mt][tap] = (hp >> 4) * 1024 + h * 256 + (hp & 15) * 16;  // + g * 512
    }
  }
  const int wb = (j * 2 + h) * 16;

  f32x16 acc[NT][2];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[t][mt][r] = 0.f;

  // ---- DMA side.  Block q of this wave = block wave + 4q of the patch; lane L fetches pixel 16*block + (L & 15), octet L >> 4.
  int item_hy[NQ], item_hx[NQ];
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int blk = (wave + WAVES * q < NBLK) ? wave + WAVES * q : wave;  // (see dma_chunk: a block past the patch repeats block `wave`)
    const int hp = blk * 16 + (lane & 15);
    item_hy[q] = hp / HWp;
    item_hx[q] = hp - item_hy[q] * HWp;
    if (hp >= NPIX) item_hy[q] = -0x10000;  // past the patch: always out of range (zeros)
  }
  unsigned voff[NQ];  // byte offset of the item's pixel + octet inside a tensor of the launch's input geometry, or out of range
  int p_unit = 0, p_s = 0, p_c0 = 0, p_chunk = 0;
  int p_wimg = 0;  // byte offset of the cursor's column group inside the weight image
  long p_patch = -1;
  bool p_same_patch = false;  // the cursor's unit reads the pixel patch of the unit before it (next column group)
  auto prefetch_unit = [&]() {
    const UnitGeom& g = p_ug;
    p_wimg = g.group * NT * a.n_chunks * IMG;
    p_same_patch = g.patch == p_patch;
    if (p_same_patch) return;  // the offsets stand
    p_patch = g.patch;
    const unsigned img0 = static_cast<unsigned>(g.n) * static_cast<unsigned>(da.img_pitch);
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int y = g.ty0 + item_hy[q] - HALO, x = g.tx0 + item_hx[q] - HALO;
      // bitwise on purpose (a short-circuit becomes a divergent branch that drags the scalar cursor into vector registers)
      const bool inside = (static_cast<unsigned>(y) < static_cast<unsigned>(d.H)) & (static_cast<unsigned>(x) < static_cast<unsigned>(d.W));
      const unsigned off = img0 + static_cast<unsigned>(y) * static_cast<unsigned>(da.row_pitch) +
                           static_cast<unsigned>(x) * static_cast<unsigned>(da.col_pitch) + static_cast<unsigned>(lane >> 4) * 16u;
      voff[q] = inside ? off : kDmaOutOfRange;
    }
  };
  // Issue the DMA of the chunk under the prefetch cursor into buffer `buf`.  Straight-line code: a wave whose block index
  // runs past the patch (22 blocks over 4 waves) fetches its first block once more (same bytes to the same place) instead
  // of branching.  Both operands come through buffer resources with a per-lane CONSTANT vector offset and everything
  // that changes in scalar registers (hipcc waits for vmcnt(0) before it rewrites an address register pair of an
  // LDS-DMA in flight, which a 64-bit vector address per weight piece needs).
  // LDS: input buffers 0 / 1, then weight buffers 0 / 1 (toggled separately: see the pipeline below)
  const int lane16 = lane * 16;
  auto dma_weights = [&](int w_slot, int group_off, int chunk) {  // the NT images of (column group, chunk) -> region slot
    unsigned char* w_dst = smem + 2 * IN_BYTES + w_slot * W_BYTES;
    const __amdgpu_buffer_rsrc_t wrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(d.weight_image), 0, da.wimg_bytes, 0x00020000);
    const int wchunk = group_off + chunk * IMG;
#pragma unroll
    for (int q = 0; q < NWQ; ++q) {
      // 1 KB block of the NT images (uniform).  A wave whose block index runs past the slot repeats a block INSIDE it.
      // Round 5: this read `: wave`, which is only inside the slot when WBLK >= WAVES.  A pointwise launch into ONE column
      // tile has WBLK = 2: waves 2 and 3 then sent blocks 2 and 3 -- source past the image (the resource returns zeros),
      // destination the OTHER weight slot, i.e. the one the MFMAs of the running chunk read from -- whenever the target was
      // slot 0.  The zeros normally land after those reads (a DMA takes longer than the two fragment reads of a pointwise
      // chunk), so the launch was right ~99.9 % of the time and dropped a chunk's weights in the rest: the eager step that
      // disagreed with the graph replay in the driver's round-4 run (tools/probes/determinism_probe.py found the launch).
      const int blk = (wave + WAVES * q < WBLK) ? wave + WAVES * q : wave % WBLK;
      static_assert(WBLK >= 1, "a slot holds at least one block");
      const int t = blk / (IMG / 1024), r = blk - t * (IMG / 1024);
      __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, (dma_lptr_t)(w_dst + blk * 1024), 16, lane16,
                                               wchunk + t * a.n_chunks * IMG + r * 1024, 0, 0);
    }
  };
  auto dma_chunk = [&](int in_buf, int w_buf, bool need_in, bool need_w) {
    unsigned char* in_dst = smem + in_buf * IN_BYTES;
    if constexpr (dma_exp::kNoInDma) need_in = false;  // (ablation builds: dma_experiments.h)
    if (need_in) {  // uniform
    const unetpp_view& V = d.in[p_s];
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(V.ptr), 0, da.view_bytes, 0x00020000);
    const int soff = ((V.oy * V.Ws + V.ox) * V.C + V.c_off + p_c0) * 2;  // phase origin, channel slice, chunk (bytes)
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int blk = (wave + WAVES * q < NBLK) ? wave + WAVES * q : wave;  // uniform
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (dma_lptr_t)(in_dst + blk * 1024), 16, static_cast<int>(voff[q]), soff, 0, 0);
    }
    }
    if constexpr (dma_exp::kNoWDma) need_w = false;
    if constexpr (dma_exp::kHalfWDma) need_w = need_w && (p_chunk & 1) == 0;
    if (need_w) dma_weights(w_buf, p_wimg, p_chunk);
  };
  // next chunk of the unit, or chunk 0 of the next unit; false when nothing is left
  auto advance = [&]() -> bool {
    if (p_chunk + 1 < a.n_chunks) {
      ++p_chunk;
      p_c0 += DKC;
      if (p_c0 >= d.in[p_s].c_len) {
        ++p_s;
        p_c0 = 0;
      }
      return true;
    }
    if (p_unit + 1 < my_units) {
      ++p_unit;
      p_chunk = 0;
      p_s = 0;
      p_c0 = 0;
      step_unit(p_ug, p_index);
      prefetch_unit();
      return true;
    }
    return false;
  };

  // The fragment reads are inline asm: hipcc's wait-count pass cannot tell that a ds_read of buffer c & 1 does not alias
  // the LDS-DMA just issued into the other buffer and would put an s_waitcnt vmcnt(0) in front of the first read of
  // every chunk (the DMA would then never overlap the MFMAs).  Reads of step s+1 are issued before the MFMAs of step s
  // and collected by one wait behind them; scheduling fences keep that order.
  struct Frag {
    u32x4 b[NT], a0, a1;
  };
  auto issue_frag = [&](auto sc, Frag& f, unsigned in_base, unsigned w_base) {  // step = tap * 2 + g: channels [16g, 16g + 16) at one tap
    constexpr int step = decltype(sc)::v, tap = step >> 1, g = step & 1;
    const unsigned wa = w_base + static_cast<unsigned>(wb);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      if (t == 0) asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.b[0]) : "v"(wa), "n"(step * DSTEP));
      else asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.b[NT - 1]) : "v"(wa), "n"(IMG + step * DSTEP));
    }
    const unsigned aa0 = in_base + static_cast<unsigned>(a_off[0][tap]), aa1 = in_base + static_cast<unsigned>(a_off[1][tap]);
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.a0) : "v"(aa0), "n"(g * 512));
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(f.a1) : "v"(aa1), "n"(g * 512));
  };
  auto wait_frag = [&](Frag& f) {
    if constexpr (NT == 1) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.b[0]), "+v"(f.a0), "+v"(f.a1));
    else asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(f.b[0]), "+v"(f.b[NT - 1]), "+v"(f.a0), "+v"(f.a1));
  };
  static_assert(NT <= 2, "fragment reads are written for one or two column tiles");

  // ---- BatchNorm-statistics epilogue (pixel-major accumulators).  Register r of acc[t][mt] of lane (j, h): pixel
  // 64*wave + 32*mt + 4h + c(r), c(r) = (r&3) + 8*(r>>2), column j.  Scratch: the input buffer just computed from.
  auto epilogue_stats = [&](unsigned char* scratch_bytes) {
    const UnitGeom& g = c_ug;
    const bool interior = (g.ty0 + TH <= d.H) && (g.tx0 + TW <= d.W);
    float* stat_lds = reinterpret_cast<float*>(scratch_bytes + 4 * 4096);  // [4 waves][32 columns][2]
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
      float* scratch = reinterpret_cast<float*>(scratch_bytes) + wave * 1024;  // [32 pixels][32 columns] fp32
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
        if (h == 0) {
          stat_lds[(wave * 32 + j) * 2 + 0] = s1;
          stat_lds[(wave * 32 + j) * 2 + 1] = s2sum;
        }
        __syncthreads();
        if (tid < tc.n_cnt) {
          float t1 = 0.f, t2 = 0.f;
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            t1 += stat_lds[(w * 32 + tid) * 2 + 0];
            t2 += stat_lds[(w * 32 + tid) * 2 + 1];
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
  // (r & 3) + 8 * (r >> 2) + 4 * h of pixel 64 * wave + 32 * mt + j.
  // Everything the epilogue READS from memory -- bias, and for the input-gradient launches the ReLU gate and the value
  // accumulated so far -- is requested BEFORE the MFMAs of the unit's last chunk (fetch_epilogue_operands) and collected
  // by the one vmcnt wait behind them, so the epilogue itself only computes and stores.  The requests are inline asm:
  // with compiler-visible loads in the loop hipcc's wait-count pass puts an s_waitcnt vmcnt(0) at the loop head (a
  // register of the load is rewritten there), which makes every unit wait for the previous unit's output stores.
  struct EpiOps {
    f32x4 b4[NT][4];
    u32x4 gate_raw[RMW ? NT : 1][2][2], old_raw[RMW ? NT : 1][2][2];   // (!RMW: never touched, no registers)
  };
  EpiOps eo;
  auto asm_load16 = [](auto& dst, const void* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(p)); };
  auto fetch_epilogue_operands = [&]() {
    const UnitGeom& g = c_ug;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const TileCols tc = decode_tile(a, g.group * NT + t);
      const unetpp_view& O = d.out[tc.ov];
      const bf16_t* optr = reinterpret_cast<const bf16_t*>(O.ptr);
      const bf16_t* gptr = reinterpret_cast<const bf16_t*>(O.gate);
      const unsigned row_stride = static_cast<unsigned>(O.sy) * O.Ws * O.C, col_stride = static_cast<unsigned>(O.sx) * O.C;
      const unsigned tile_base = view_pixel_offset32(O, g.n, g.ty0, g.tx0) + tc.nt * 32;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        eo.b4[t][q] = f32x4{0.f, 0.f, 0.f, 0.f};
        if (d.bias != nullptr) {  // uniform; columns past the tile read column 0's (valid address, value unused)
          const int c = (8 * q + 4 * h < tc.n_cnt) ? 8 * q + 4 * h : 0;
          asm_load16(eo.b4[t][q], d.bias + tc.n0 + c);
        }
      }
      if (RMW && (gptr != nullptr || O.accumulate)) {  // uniform
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
          const bool pix_ok = (g.ty0 + epi_py[mt] < d.H) && (g.tx0 + epi_px[mt] < d.W);
          const unsigned pbase = tile_base + static_cast<unsigned>(epi_py[mt]) * row_stride + static_cast<unsigned>(epi_px[mt]) * col_stride;
#pragma unroll
          for (int half = 0; half < 2; ++half) {
            const int c0 = 16 * half + 8 * (1 - h);
            const unsigned off = (pix_ok && c0 < tc.n_cnt) ? pbase + c0 : tile_base;  // dead pieces: any valid address
            if (gptr != nullptr) asm_load16(eo.gate_raw[RMW ? t : 0][mt][half], gptr + off);
            if (O.accumulate) asm_load16(eo.old_raw[RMW ? t : 0][mt][half], optr + off);
          }
        }
      }
    }
  };
  auto epilogue_direct = [&]() {
    const UnitGeom& g = c_ug;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      const TileCols tc = decode_tile(a, g.group * NT + t);
      const unetpp_view& O = d.out[tc.ov];
      bf16_t* optr = reinterpret_cast<bf16_t*>(O.ptr);
      const bool has_gate = RMW && O.gate != nullptr, acc_out = RMW && O.accumulate != 0;  // uniform
      const unsigned row_stride = static_cast<unsigned>(O.sy) * O.Ws * O.C, col_stride = static_cast<unsigned>(O.sx) * O.C;
      const unsigned tile_base = view_pixel_offset32(O, g.n, g.ty0, g.tx0) + tc.nt * 32;
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int py = epi_py[mt], px = epi_px[mt];
        const bool pix_ok = (g.ty0 + py < d.H) && (g.tx0 + px < d.W);
        const unsigned pbase = tile_base + static_cast<unsigned>(py) * row_stride + static_cast<unsigned>(px) * col_stride;
        unsigned pk[4][2];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          float v[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            v[e] = acc[t][mt][4 * q + e] + ((8 * q + 4 * h < tc.n_cnt) ? eo.b4[t][q][e] : 0.f);
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
          if (has_gate || acc_out) {
            float v[8];
            unpack8(out, v);
            float gt[8];
            if (has_gate) unpack8(eo.gate_raw[RMW ? t : 0][mt][half], gt);
            if (has_gate && !O.gate_sum) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (gt[e] > 0.f) ? v[e] : 0.f;
            }
            if (acc_out) {
              float old[8];
              unpack8(eo.old_raw[RMW ? t : 0][mt][half], old);
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] += old[e];
            }
            if (has_gate && O.gate_sum) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = (gt[e] > 0.f) ? v[e] : 0.f;
            }
            out = pack8(v);
          }
          if constexpr (dma_exp::kNoStore) {
            asm volatile("" ::"v"(out), "v"(optr + pbase + c0));
          } else if constexpr (dma_exp::kStoreLinear) {
            if (pix_ok && c0 < tc.n_cnt) *reinterpret_cast<u32x4*>(optr + tile_base + ((wave * 4 + mt * 2 + half) * 64 + lane) * 8) = out;
          } else {
            if (pix_ok && c0 < tc.n_cnt) *reinterpret_cast<u32x4*>(optr + pbase + c0) = out;
          }
        }
      }
    }
  };

  // ---- pipeline: a chunk is computed from input buffer in_cur / weight buffer w_cur while the DMA of the next chunk fills
  // the other ones.  Per chunk: issue DMA(next); MFMAs; wait for the DMA (before the epilogue's stores are issued: vmcnt
  // counts in order); epilogue of a finished unit; one barrier (every wave's share of the next chunk has landed, nobody
  // reads the current buffers any more, the epilogue's scratch is free).
  // What is NOT fetched again: the weight image when the whole launch uses one (a single chunk and a single column group:
  // the 32 -> 32 layers of level 0) -- it stays in its buffer; the input patch when the next unit is the next column
  // group of the same patch and a unit is one chunk (input gradients of a 32-channel dy into 64..128 channels) -- the
  // input buffer is not toggled.  The kernel is bound by what a CU's memory pipeline moves (~10 B/clk), not by HBM. ----
  // uniform: the launch's whole weight image fits the weight region (slot = column group * chunks + chunk)
  const bool w_resident = a.n_groups * a.n_chunks <= W_SLOTS;
  const bool in_reuse = !STATS && a.n_chunks == 1;              // (the statistics epilogue uses the input buffer as scratch)
  prefetch_unit();
  if (w_resident) {
    for (int g = 0; g < a.n_groups; ++g)
      for (int c = 0; c < a.n_chunks; ++c) dma_weights(g * a.n_chunks + c, g * NT * a.n_chunks * IMG, c);  // for the whole launch
  }
  dma_chunk(0, 0, true, !w_resident);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();

  int c_chunk = 0, in_cur = 0, w_cur = 0;
  bool more = advance();
#ifdef UNETPP_DMA_STAMPS
  unsigned long long st_acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long st_last = __builtin_amdgcn_s_memtime();
  unsigned long long n_chunks_done = 0;
#endif
  while (true) {
    DMA_STAMP(0);  // 0: cursor advance (end of the previous iteration)
    const bool need_in = more && !(in_reuse && p_same_patch), need_w = more && !w_resident;
    if (more) dma_chunk(in_cur ^ 1, w_cur ^ 1, need_in, need_w);
    if constexpr (!STATS) {
      if (c_chunk + 1 == a.n_chunks) fetch_epilogue_operands();  // uniform
    }
    DMA_STAMP(1);  // 1: DMA issue + epilogue operand requests
    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
    const unsigned in_base = lds0 + static_cast<unsigned>(in_cur) * IN_BYTES;
    const unsigned w_base = lds0 + 2 * IN_BYTES + static_cast<unsigned>(w_resident ? c_ug.group * a.n_chunks + c_chunk : w_cur) * W_BYTES;
    Frag fr[2];
    asm volatile("" ::: "memory");  // the reads below stay behind the barrier that published this buffer
    issue_frag(IC<0>{}, fr[0], in_base, w_base);
    wait_frag(fr[0]);
    static_for<TAPS * 2>([&](auto sc) {
      constexpr int step = decltype(sc)::v, cs = step & 1, ns = cs ^ 1;
      if constexpr (step + 1 < TAPS * 2) issue_frag(IC<step + 1>{}, fr[ns], in_base, w_base);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int ct = 0; ct < (dma_exp::kNoMfma ? 0 : NT); ++ct) {
        if constexpr (STATS) {
          acc[ct][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[cs].a0),
                                                               __builtin_bit_cast(bf16x8, fr[cs].b[ct]), acc[ct][0], 0, 0, 0);
          acc[ct][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[cs].a1),
                                                               __builtin_bit_cast(bf16x8, fr[cs].b[ct]), acc[ct][1], 0, 0, 0);
        } else {  // swapped roles: rows = output columns, columns = pixels (same fragments: see the lane maps)
          acc[ct][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[cs].b[ct]),
                                                               __builtin_bit_cast(bf16x8, fr[cs].a0), acc[ct][0], 0, 0, 0);
          acc[ct][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fr[cs].b[ct]),
                                                               __builtin_bit_cast(bf16x8, fr[cs].a1), acc[ct][1], 0, 0, 0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (step + 1 < TAPS * 2) wait_frag(fr[ns]);
    });
    DMA_STAMP(2);  // 2: LDS fragment reads + MFMAs
    // this wave's share of chunk c+1 has landed, and the epilogue's operands (requested before the MFMAs) are in
    if constexpr (STATS) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        asm volatile("s_waitcnt vmcnt(0)" : "+v"(eo.b4[t][0]), "+v"(eo.b4[t][1]), "+v"(eo.b4[t][2]), "+v"(eo.b4[t][3])::"memory");
        if constexpr (RMW) {
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(eo.gate_raw[t][0][0]), "+v"(eo.gate_raw[t][0][1]), "+v"(eo.gate_raw[t][1][0]), "+v"(eo.gate_raw[t][1][1])::"memory");
          asm volatile("s_waitcnt vmcnt(0)" : "+v"(eo.old_raw[t][0][0]), "+v"(eo.old_raw[t][0][1]), "+v"(eo.old_raw[t][1][0]), "+v"(eo.old_raw[t][1][1])::"memory");
        }
      }
    }
    DMA_STAMP(3);  // 3: wait for the DMA (and whatever is older: the previous unit's stores)
#ifdef UNETPP_DMA_STAMPS
    ++n_chunks_done;
#endif
    if (c_chunk + 1 == a.n_chunks) {
      if constexpr (STATS) {
        __syncthreads();  // the transposing epilogue uses the buffer just computed from as scratch: all waves are done with it
        epilogue_stats(smem + in_cur * IN_BYTES);
      } else {
        if constexpr (!dma_exp::kNoEpi) {
          epilogue_direct();
        } else {  // The accumulators must stay live: until round 5 this variant dropped the call and nothing else, hipcc
          // then deleted every MFMA of the kernel as dead code, and the "epilogue share" read off it (33-48 %) was the
          // epilogue AND the matrix phase (profiles/r5/ablation_gemm_bf16_dma_stores_c5.txt has the corrected table)
#pragma unroll
          for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
              asm volatile("" : "+v"(acc[t][mt]));
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[t][mt][r] = 0.f;
            }
        }
      }
      step_unit(c_ug, c_index);
      c_chunk = 0;
    } else {
      ++c_chunk;
    }
    DMA_STAMP(4);  // 4: epilogue (arithmetic + store issue)
    if (!more) break;
    // raw barrier: __syncthreads() carries a fence that drains vmcnt(0), i.e. would wait for this unit's output stores.
    // What the barrier has to order is already complete in every wave: its fragment reads of buffer `cur` (collected by
    // the lgkmcnt waits of the MFMA phase), its share of the next chunk's DMA (the vmcnt wait above, issued before the
    // stores), and -- statistics launches -- the epilogue's LDS traffic (it ends in a __syncthreads of its own).
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    DMA_STAMP(5);  // 5: barrier (= the MFMA phase of the other wave of this SIMD, mostly)
    if (need_in) in_cur ^= 1;
    if (need_w) w_cur ^= 1;
    more = advance();
  }
#ifdef UNETPP_DMA_STAMPS
  if (tid == 0) {
    for (int i = 0; i < 8; ++i) atomicAdd(&g_dma_stamps[i], st_acc[i]);
    atomicAdd(&g_dma_stamps[8], n_chunks_done);
    atomicAdd(&g_dma_stamps[9], static_cast<unsigned long long>(my_units));
    atomicAdd(&g_dma_stamps[10], 1ull);
  }
#endif
}

bool dma_env_off() { return opt_value(OPT_BF16_NO_DMA, 0) != 0; }  // unetpp_debug_set: tests compare the two kernels

}  // namespace

#ifdef UNETPP_DMA_STAMPS
extern "C" int unetpp_debug_dma_stamps(unsigned long long* out16, int reset) {  // profiling builds only
  if (out16 != nullptr && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_dma_stamps), sizeof(g_dma_stamps)) != hipSuccess)
    return UNETPP_ELAUNCH;
  if (reset) {
    unsigned long long zero[16] = {};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_dma_stamps), zero, sizeof(zero)) != hipSuccess) return UNETPP_ELAUNCH;
  }
  return UNETPP_OK;
}
#endif

// UNETPP_OK after launching; 1 when the descriptor is not one this kernel takes (the caller falls back to gemm_bf16.hip)
int launch_gemm_bf16_dma(const unetpp_gemm_desc* d, hipStream_t st) {
  if (dma_env_off()) return 1;
  DmaArgs da;
  FastArgs& a = da.f;
  if (!bf16_gemm_args(d, a) || d->weight_image == nullptr) return 1;
  if (d->stats_partial != nullptr && d->n_out != 1) return 1;
  // Which form (tools/bench_kernels.py, 512 x 512 x 8; UNETPP_BF16_DMA_FORM = 0 / 4 / 8 forces none / one of them, and
  // UNETPP_BF16_DMA_ALL=1 lifts the restrictions: the tests run every shape class through both kernels):
  //   8 waves, 512-pixel patches   plain 3x3 launches on images at least 32 wide with more than one chunk or column tile
  //   4 waves, 256-pixel patches   3x3 launches of ONE chunk into ONE tile (32 -> 32: its image is resident either way
  //                                and two independent workgroups per CU overlap better); since round 4 also: plain 3x3
  //                                launches too small for the 8-wave form, BatchNorm-statistics launches with plain inputs,
  //                                the pointwise GEMMs (see the switches below)
  //   neither (gemm_bf16.hip)      load transforms (BatchNorm fold / ReLU / gate on load), unaligned channel slices
  const bool all = opt_value(OPT_BF16_DMA_ALL, 0) != 0;
  const bool stats = d->stats_partial != nullptr;
  int form = 4;
  const int cus = device_cu_count();
  if (cus <= 0) return UNETPP_ELAUNCH;
  // (the 8-wave form only when its 512-pixel units still cover the chip: the deepest layers of a small image do not)
  const long units8 = static_cast<long>(d->N) * ((d->H + 15) / 16) * a.tiles_x * ((a.n_tiles % 2 == 0) ? a.n_tiles / 2 : a.n_tiles);
  const long min8 = opt_value(OPT_BF16_DMA_MIN8, 2);  // (A/B knob)
  // (one-chunk one-tile launches in the 8-wave form as well: 0 .. +2 % on the configs[3] step depending on the box, not taken)
  if (d->taps == 9 && !stats && a.log2tw == 5 && (a.n_chunks > 1 || a.n_tiles > 1) && units8 >= min8 * cus) form = 8;
  if (opt_is_set(OPT_BF16_DMA_FORM)) {
    const int want = static_cast<int>(opt_value(OPT_BF16_DMA_FORM, 4));
    if (want == 0) return 1;
    if (want == 4 || (want == 8 && d->taps == 9 && !stats && a.log2tw == 5)) form = want;
  }
  // Round 4: plain 3x3 launches too small for the 8-wave form (the deep levels: 24 x 24 .. 96 x 96 images) take the 4-wave
  // form as well instead of falling back to the register kernel: its 256-pixel x 32-column units are the finest the
  // library has, and at these sizes the number of units per CU is what counts (288 / 144 / 80 units of the 8-wave shape
  // for 256 CUs at levels 2-4 of configs[4]).  UNETPP_BF16_DMA_SMALL=0 restores the round-3 choice (A/B runs).
  const bool small_too = opt_value(OPT_BF16_DMA_SMALL, 1) != 0;
  const bool small3x3 = small_too && d->taps == 9 && !stats && form == 4 && a.n_tiles > 1;
  // BatchNorm-statistics launches with plain inputs (conv1 of the encoder levels >= 1) through the 4-wave statistics
  // instantiation: +0.3 % on the configs[4] step, nothing at configs[3] (same box, alternating); =0 switches it off
  const bool stats_too = opt_value(OPT_BF16_DMA_STATS, 1) != 0;
  const bool stats3x3 = stats_too && d->taps == 9 && stats && form == 4 && d->n_out == 1;
  // the transposed-convolution GEMMs (pointwise, four phase views) through this kernel as well: +1 % on the configs[3]
  // step, nothing at configs[4] (same box, alternating); UNETPP_BF16_DMA_POINTWISE=0 keeps them on the register kernel
  const bool pointwise = opt_value(OPT_BF16_DMA_POINTWISE, 1) != 0 && d->taps == 1 && !stats;
  if (!all && !small3x3 && !stats3x3 && !pointwise && form == 4 && (d->taps != 9 || a.n_tiles != 1 || stats)) return 1;
  const unetpp_view& V0 = d->in[0];
  for (int i = 0; i < d->n_in; ++i) {
    const unetpp_view& v = d->in[i];
    if (v.scale != nullptr || v.relu != 0 || v.gate != nullptr) return 1;       // load transforms need VALU on the way
    if ((v.c_len % DKC) != 0) return 1;                                          // whole 32-channel chunks
    if (v.C != V0.C || v.Hs != V0.Hs || v.Ws != V0.Ws || v.sy != V0.sy || v.sx != V0.sx) return 1;  // one geometry
    if (d->taps == 9 && (v.sy != 1 || v.sx != 1 || v.oy != 0 || v.ox != 0 || v.Hs != d->H || v.Ws != d->W)) return 1;
  }
  const long view_bytes = static_cast<long>(d->N) * V0.Hs * V0.Ws * V0.C * 2;
  if (view_bytes > 0x7fffffffL) return 1;
  da.pitch = V0.C * 2;
  da.row_pitch = V0.sy * V0.Ws * V0.C * 2;
  da.col_pitch = V0.sx * V0.C * 2;
  da.img_pitch = V0.Hs * V0.Ws * V0.C * 2;
  da.view_bytes = static_cast<int>(view_bytes);
  const long wimg_bytes = static_cast<long>(a.n_tiles) * a.n_chunks * (d->taps * 2 * DSTEP);
  if (wimg_bytes > 0x7fffffffL) return 1;
  da.wimg_bytes = static_cast<int>(wimg_bytes);
  da.unit_base = 0;
  if (form == 8) {
    // 16 x 32 patches; two column tiles per unit when the launch has an even number of them
    const int nt = (a.n_tiles % 2 == 0) ? 2 : 1;
    a.nt_unit = nt;
    a.n_groups = a.n_tiles / nt;
    a.tiles_y = (d->H + 15) / 16;
    a.total_blocks = static_cast<long>(d->N) * a.tiles_y * a.tiles_x * a.n_groups;
    long workers = static_cast<long>(cus) & ~7L;  // one 8-wave workgroup per CU
    if (workers < 8) workers = 8;
    // Round 4: the tail.  Units are equal, every workgroup walks ceil(units / workers) of them, so 1152 units on 256 CUs
    // (level 0 of configs[4]: 384 = 3 * 128) cost 5 rounds for 4.5 rounds of work, 576 units 3 for 2.25.  The whole rounds
    // stay here; the patch ROWS behind them go to a second launch of the 4-wave form, whose 256-pixel x 32-column units are
    // a quarter of the work each (a 16-row patch row here = two 8-row patch rows there, so the remainder is a contiguous
    // unit range of that form as long as the split sits on a patch-row boundary and H is a multiple of 16).
    // MEASURED AND LEFT OFF (UNETPP_BF16_DMA_SPLIT=1 turns it on; read per launch: the tests run it): configs[4] 10.41 ->
    // 10.53 ms per step on one box, alternating.  Half (level 0) or a quarter (level 1) of the workgroups carry the extra
    // unit, the others idle for ~10 % / ~25 % of the launch -- but the second launch's gap, prologue and weight re-staging
    // cost more than that idle time gives back.
    const bool split_tail = opt_value(OPT_BF16_DMA_SPLIT, 0) == 1;
    long units_here = a.total_blocks;
    const long row8 = static_cast<long>(a.tiles_x) * a.n_groups;            // units of one 16-row patch row
    if (split_tail && (d->H % 16) == 0 && a.total_blocks > workers) {
      const long whole = (a.total_blocks / workers) * workers;
      const long rows_here = whole / row8;
      const long rest8 = a.total_blocks - rows_here * row8;                  // units of this form that move
      const long units4 = rest8 / a.n_groups * 2 * a.n_tiles;                // two 8-row patches per patch, one tile per unit
      const long workers4 = (2L * cus) & ~7L;
      // cost in rounds of THIS form's unit time; a unit of the other form is a quarter of the work, priced at 0.35
      const double now = static_cast<double>((a.total_blocks + workers - 1) / workers);
      const double split = static_cast<double>((rows_here * row8 + workers - 1) / workers) +
                           0.35 * static_cast<double>((units4 + workers4 - 1) / workers4);
      if (rows_here > 0 && rest8 > 0 && split < 0.95 * now) units_here = rows_here * row8;
    }
    const long all_units = a.total_blocks;
    a.total_blocks = units_here;
    {
      const dim3 grid(static_cast<unsigned>(a.total_blocks <= workers ? a.total_blocks : workers)), block(512);
      if (nt == 2) hipLaunchKernelGGL((gemm_bf16_dma_kernel<9, 5, 2, false, 8>), grid, block, 0, st, da);
      else hipLaunchKernelGGL((gemm_bf16_dma_kernel<9, 5, 1, false, 8>), grid, block, 0, st, da);
    }
    if (units_here < all_units) {  // the tail: 8-row patches from patch row 2 * rows_here on, one column tile per unit
      const long rows_here = units_here / row8;
      a.nt_unit = 1;
      a.n_groups = a.n_tiles;
      a.tiles_y = (d->H + 7) / 8;
      const long row4 = static_cast<long>(a.tiles_x) * a.n_groups;
      const long total4 = static_cast<long>(d->N) * a.tiles_y * row4;
      da.unit_base = static_cast<int>(2 * rows_here * row4);
      a.total_blocks = total4 - da.unit_base;
      long workers4 = (2L * cus) & ~7L;
      if (workers4 < 8) workers4 = 8;
      const dim3 grid(static_cast<unsigned>(a.total_blocks <= workers4 ? a.total_blocks : workers4)), block(kThreads);
      hipLaunchKernelGGL((gemm_bf16_dma_kernel<9, 5, 1, false>), grid, block, 0, st, da);
    }
    note_kernel("gemm_bf16_dma_kernel<9>");
    return launch_status();
  }
  // 4 waves, 3x3: one column tile per unit (two weight buffers of 18 KB beside two input buffers of 22 KB: 80 KB, two
  // workgroups per CU); pointwise: two tiles when the launch has an even number of them
  if (d->taps == 9 && a.nt_unit == 2) {
    a.nt_unit = 1;
    a.n_groups = a.n_tiles;
    a.total_blocks *= 2;
  }
  // pointwise launches whose output views carry no gate / accumulate flag (the forward of a transposed or 1x1 convolution):
  // the three-per-CU instantiation (UNETPP_BF16_PW_PLAIN=0 / unetpp_debug_set keeps them on the two-per-CU one: A/B runs)
  bool plain_out = d->taps == 1 && !stats && opt_value(OPT_BF16_PW_PLAIN, 1) != 0;
  for (int i = 0; i < d->n_out; ++i) plain_out = plain_out && d->out[i].gate == nullptr && d->out[i].accumulate == 0;
  long workers = ((plain_out ? 3L : 2L) * cus) & ~7L;
  if (workers < 8) workers = 8;
  const dim3 grid(static_cast<unsigned>(a.total_blocks <= workers ? a.total_blocks : workers)), block(kThreads);
#define UNETPP_LAUNCH_BF16_DMA(T, NTU, ST, RW)                                                                         \
  do {                                                                                                                 \
    if (a.log2tw == 5) hipLaunchKernelGGL((gemm_bf16_dma_kernel<T, 5, NTU, ST, 4, RW>), grid, block, 0, st, da);      \
    else if (a.log2tw == 4) hipLaunchKernelGGL((gemm_bf16_dma_kernel<T, 4, NTU, ST, 4, RW>), grid, block, 0, st, da); \
    else hipLaunchKernelGGL((gemm_bf16_dma_kernel<T, 3, NTU, ST, 4, RW>), grid, block, 0, st, da);                    \
  } while (0)
  if (d->taps == 9) {
    if (stats) UNETPP_LAUNCH_BF16_DMA(9, 1, true, true);
    else UNETPP_LAUNCH_BF16_DMA(9, 1, false, true);
  } else if (a.nt_unit == 2) {
    if (plain_out) UNETPP_LAUNCH_BF16_DMA(1, 2, false, false);
    else UNETPP_LAUNCH_BF16_DMA(1, 2, false, true);
  } else {
    if (plain_out) UNETPP_LAUNCH_BF16_DMA(1, 1, false, false);
    else UNETPP_LAUNCH_BF16_DMA(1, 1, false, true);
  }
#undef UNETPP_LAUNCH_BF16_DMA
  note_kernel(d->taps == 9 ? "gemm_bf16_dma_kernel<9>" : "gemm_bf16_dma_kernel<1>");
  return launch_status();
}

}  // namespace unetpp

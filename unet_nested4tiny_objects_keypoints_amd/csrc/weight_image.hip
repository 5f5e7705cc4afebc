// Weight images of the fast GEMM kernels, built straight from the torch-layout parameters.
//
// A fast kernel reads its B operand as an LDS image per (column tile, K chunk): the direct kernels
// (gemm_fast.hip) [tap][g 2][col 32][half' 2][4] over 16-channel chunks, the Winograd kernel (gemm_wino.hip)
// [s 2][xi 16][g 4][col 16][nh 2] of U = G g G^T over 8-channel chunks.  The element of the GEMM weight that lands in
// an image slot is W[tap][k][n]; `unetpp_weight_src` says where that element sits in memory:
//     src[tap' * s_t + (k % k_inner) * s_k + (k / k_inner) * s_ko + (n % n_inner) * s_n + (n / n_inner) * s_no],
// tap' = flip ? taps - 1 - tap : tap, which covers the packed [taps][K][N] operand as well as nn.Conv2d
// ([co,ci,3,3]: forward and 180-degree-rotated input-gradient form) and nn.ConvTranspose2d ([ci,co,2,2]: forward
// with N = 4 phases x co, input gradient with K = 4 phases x co) parameters -- no intermediate re-layout launch.
// unetpp_gemm_pack_weight_images builds the images of many launches (a whole forward or backward pass) in ONE launch
// from a table of jobs in device memory.
#include "bf16_common.h"
#include "common.h"
#include "gemm_units.h"

namespace unetpp {
namespace {

constexpr int kKindFast = 0, kKindWino = 1, kKindBf16 = 2;  // bf16: gemm_bf16.hip, [tap][g 2][col 32][h 2][8 bf16], 32-channel chunks

struct PackGeom {  // what the image layout depends on: the channel structure of the launch
  int kind, taps, kc, ncol;  // kc = channels per K chunk, ncol = columns per tile
  int n_in, n_out;
  int in_len[UNETPP_MAX_VIEWS], out_len[UNETPP_MAX_VIEWS];
  int n_chunks, n_tiles;
  long floats;
};

__host__ __device__ inline long image_floats_per_tile_chunk(int kind, int taps) { return kind == kKindWino ? 4096 : taps * 512L; }

__host__ __device__ inline void finish_geom(PackGeom& g) {
  g.n_chunks = 0;
  g.n_tiles = 0;
  for (int i = 0; i < g.n_in; ++i) g.n_chunks += (g.in_len[i] + g.kc - 1) / g.kc;
  for (int i = 0; i < g.n_out; ++i) g.n_tiles += (g.out_len[i] + g.ncol - 1) / g.ncol;
  g.floats = static_cast<long>(g.n_tiles) * g.n_chunks * image_floats_per_tile_chunk(g.kind, g.taps);
}

__device__ __forceinline__ float wsrc_at(const unetpp_weight_src& w, int taps, int tap, int k, int n) {
  const int tt = w.flip ? taps - 1 - tap : tap;
  const int ki = w.k_inner > 0 ? k % w.k_inner : k, ko = w.k_inner > 0 ? k / w.k_inner : 0;
  const int ni = w.n_inner > 0 ? n % w.n_inner : n, no = w.n_inner > 0 ? n / w.n_inner : 0;
  return w.src[tt * w.s_t + ki * w.s_k + ko * w.s_ko + ni * w.s_n + no * w.s_no];
}

// (tile, chunk) of an image row -> input view / first GEMM row of the chunk, output view / first GEMM column of the tile
struct RowOrigin {
  int k0, k_room;  // GEMM row of the chunk's channel 0; channels left in its view from there
  int n0, n_room;  // GEMM column of the tile's column 0; columns left in its view from there
};
__device__ __forceinline__ RowOrigin row_origin(const PackGeom& g, int nt, int chunk) {
  int kbase = 0, v = 0;
  for (; v < g.n_in - 1; ++v) {
    const int ch = (g.in_len[v] + g.kc - 1) / g.kc;
    if (chunk < ch) break;
    chunk -= ch;
    kbase += g.in_len[v];
  }
  int col_base = 0, ov = 0;
  for (; ov < g.n_out - 1; ++ov) {
    const int tv = (g.out_len[ov] + g.ncol - 1) / g.ncol;
    if (nt < tv) break;
    nt -= tv;
    col_base += g.out_len[ov];
  }
  RowOrigin o;
  o.k0 = kbase + chunk * g.kc;
  o.k_room = g.in_len[v] - chunk * g.kc;
  o.n0 = col_base + nt * g.ncol;
  o.n_room = g.out_len[ov] - nt * g.ncol;
  return o;
}

// The batched pack kernel walks whole image rows: row = (tile, chunk[, tap]) is decoded once per workgroup iteration
// (uniform: scalar unit), a thread only splits its slot's bit fields.  (Decoding per element cost ~80 instructions
// per 4 bytes written: 175 us per launch for the 73 MB of bf16 images of the depth-5 network.)
__device__ __forceinline__ void fill_image_row(const PackGeom& g, const unetpp_weight_src& w, long row, float* __restrict__ img) {
  const int tid = threadIdx.x;
  if (g.kind == kKindWino) {  // row = (tile, chunk): 256 (channel, column) pairs x 16 transform elements
    const RowOrigin o = row_origin(g, static_cast<int>(row / g.n_chunks), static_cast<int>(row % g.n_chunks));
    const int t = tid;  // nh | col << 1 | gq << 5 | s << 7
    const int kk = 2 * ((t >> 5) & 3) + ((t >> 7) & 1), cl = 16 * (t & 1) + ((t >> 1) & 15);
    float* dst = img + row * 4096 + (t & 127) + (static_cast<long>(t >> 7) << 11);
    if (kk >= o.k_room || cl >= o.n_room) {
#pragma unroll
      for (int xi = 0; xi < 16; ++xi) dst[xi << 7] = 0.f;
      return;
    }
    const int k = o.k0 + kk, n = o.n0 + cl;
    float tr[4][3];  // G g: row ra of G down the filter rows, per filter column
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float w0 = wsrc_at(w, 9, 0 * 3 + c, k, n), w1 = wsrc_at(w, 9, 1 * 3 + c, k, n), w2 = wsrc_at(w, 9, 2 * 3 + c, k, n);
      tr[0][c] = w0;
      tr[1][c] = 0.5f * (w0 + w1 + w2);
      tr[2][c] = 0.5f * (w0 - w1 + w2);
      tr[3][c] = w2;
    }
#pragma unroll
    for (int ra = 0; ra < 4; ++ra) {
      dst[(4 * ra + 0) << 7] = tr[ra][0];
      dst[(4 * ra + 1) << 7] = 0.5f * (tr[ra][0] + tr[ra][1] + tr[ra][2]);
      dst[(4 * ra + 2) << 7] = 0.5f * (tr[ra][0] - tr[ra][1] + tr[ra][2]);
      dst[(4 * ra + 3) << 7] = tr[ra][2];
    }
    return;
  }
  // direct kernels: row = (tile, chunk, tap), 512 four-byte slots
  const int tap = static_cast<int>(row % g.taps);
  const long rc = row / g.taps;
  const RowOrigin o = row_origin(g, static_cast<int>(rc / g.n_chunks), static_cast<int>(rc % g.n_chunks));
  float* dst = img + row * 512;
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int e = tid + q * 256;
    if (g.kind == kKindBf16) {  // slot = two bf16: elements 2e, 2e+1 of [g 2][col 32][h 2][8]
      const int i0 = 2 * e;
      const int kk = 16 * ((i0 >> 9) & 1) + 8 * ((i0 >> 3) & 1) + (i0 & 7), cl = (i0 >> 4) & 31;
      float a = 0.f, b = 0.f;
      if (cl < o.n_room) {
        if (kk < o.k_room) a = wsrc_at(w, g.taps, tap, o.k0 + kk, o.n0 + cl);
        if (kk + 1 < o.k_room) b = wsrc_at(w, g.taps, tap, o.k0 + kk + 1, o.n0 + cl);
      }
      dst[e] = __uint_as_float(pack_bf2(a, b));
    } else {  // [g 2][col 32][half' 2][4], half' = half ^ ((col>>3)&1)
      const int j = (e >> 3) & 31;
      const int kk = 8 * ((e >> 8) & 1) + 4 * (((e >> 2) & 1) ^ ((j >> 3) & 1)) + (e & 3);
      dst[e] = (kk < o.k_room && j < o.n_room) ? wsrc_at(w, g.taps, tap, o.k0 + kk, o.n0 + j) : 0.f;
    }
  }
}

// bf16 3x3 images, nine rows at a time: the 32 x 32 x 9 source block of a (tile, chunk) goes through LDS, read in the
// order it lies in memory (the taps of a (channel, column) pair are adjacent, then whichever of channel / column has
// the smaller stride) -- fill_image_row reads it as 4-byte pieces 36 bytes apart, nine times over (once per tap) --
// and leaves as 16-byte stores.
__device__ __forceinline__ void fill_bf16_3x3_block(const PackGeom& g, const unetpp_weight_src& w, long rc,
                                                     float* __restrict__ img, float* lds /* [9][32][33] */) {
  const int tid = threadIdx.x;
  const RowOrigin o = row_origin(g, static_cast<int>(rc / g.n_chunks), static_cast<int>(rc % g.n_chunks));
  const bool k_inner_most = w.s_k < w.s_n;  // forward layout [co][ci][3][3]: channel k = ci runs faster than column n = co
  __syncthreads();  // the previous block's readers are done
  // 16-byte reads where the nine taps of 32 consecutive inner indices form one aligned 1152-byte run (plain conv
  // parameters: tap stride 1, inner stride 9, 16-byte aligned block origin), 4-byte reads otherwise
  const long inner_stride = k_inner_most ? w.s_k : w.s_n, outer_stride = k_inner_most ? w.s_n : w.s_k;
  const int inner_room = k_inner_most ? o.k_room : o.n_room, outer_room = k_inner_most ? o.n_room : o.k_room;
  const long origin = static_cast<long>(o.k0) * w.s_k + static_cast<long>(o.n0) * w.s_n;
  const bool wide = w.s_t == 1 && inner_stride == 9 && w.k_inner <= 0 && w.n_inner <= 0 && inner_room >= 32 &&
                    ((reinterpret_cast<uintptr_t>(w.src + origin) | static_cast<uintptr_t>(outer_stride * 4)) & 15) == 0;
  if (wide) {
#pragma unroll 3
    for (int q = tid; q < 32 * 72; q += 256) {  // 72 float4 per outer index
      const int oo = q / 72, r4 = q - oo * 72;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (oo < outer_room) v = *reinterpret_cast<const f32x4*>(w.src + origin + oo * outer_stride + 4 * r4);
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        const int f = 4 * r4 + c, m = f / 9;
        const int tap_m = f - 9 * m, tap = w.flip ? 8 - tap_m : tap_m;
        const int kk = k_inner_most ? m : oo, cl = k_inner_most ? oo : m;
        lds[(tap * 32 + kk) * 33 + cl] = v[c];
      }
    }
  } else {
#pragma unroll 6
    for (int idx = tid; idx < 9 * 32 * 32; idx += 256) {
      const int tap = idx % 9, m = (idx / 9) & 31, oo = idx / (9 * 32);
      const int kk = k_inner_most ? m : oo, cl = k_inner_most ? oo : m;
      lds[(tap * 32 + kk) * 33 + cl] = (kk < o.k_room && cl < o.n_room) ? wsrc_at(w, 9, tap, o.k0 + kk, o.n0 + cl) : 0.f;
    }
  }
  __syncthreads();
  // one 16-byte store per thread and round: four slots = the eight channels 16 gq + 8 hs .. + 7 of one column
  f32x4* dst4 = reinterpret_cast<f32x4*>(img + rc * 9 * 512);
#pragma unroll 3
  for (int s4 = tid; s4 < 9 * 128; s4 += 256) {
    const int tap = s4 >> 7, e4 = s4 & 127;  // slots 4 e4 .. 4 e4 + 3 of the tap's row
    const int cl = (e4 >> 1) & 31, kk0 = 16 * ((e4 >> 6) & 1) + 8 * (e4 & 1);
    f32x4 out;
#pragma unroll
    for (int c = 0; c < 4; ++c)
      out[c] = __uint_as_float(pack_bf2(lds[(tap * 32 + kk0 + 2 * c) * 33 + cl], lds[(tap * 32 + kk0 + 2 * c + 1) * 33 + cl]));
    dst4[s4] = out;
  }
}

// value of image element i
__device__ __forceinline__ float image_element(const PackGeom& g, const unetpp_weight_src& w, long i) {
  int kk, cin_local, tap = 0, xi = 0;
  long r;
  if (g.kind == kKindWino) {  // [s 2][xi 16][g 4][col 16][nh 2]
    const int nh = i & 1, col = (i >> 1) & 15, gq = (i >> 5) & 3, s = (i >> 11) & 1;
    xi = (i >> 7) & 15;
    kk = 2 * gq + s;
    cin_local = 16 * nh + col;
    r = i >> 12;
  } else if (g.kind == kKindBf16) {  // i counts bf16 elements: [tap][g 2][col 32][h 2][8]
    const int e = i & 7, hs = (i >> 3) & 1, j = (i >> 4) & 31, gq = (i >> 9) & 1;
    r = i >> 10;
    tap = static_cast<int>(r % g.taps);
    r /= g.taps;
    kk = 16 * gq + 8 * hs + e;
    cin_local = j;
  } else {  // [tap][g 2][col 32][half' 2][4], half' = half ^ ((col>>3)&1)
    const int e = i & 3, hs = (i >> 2) & 1, j = (i >> 3) & 31, gq = (i >> 8) & 1;
    r = i >> 9;
    tap = static_cast<int>(r % g.taps);
    r /= g.taps;
    kk = 8 * gq + 4 * (hs ^ ((j >> 3) & 1)) + e;
    cin_local = j;
  }
  int chunk = static_cast<int>(r % g.n_chunks);
  int nt = static_cast<int>(r / g.n_chunks);
  int kbase = 0, v = 0;
  for (; v < g.n_in - 1; ++v) {
    const int ch = (g.in_len[v] + g.kc - 1) / g.kc;
    if (chunk < ch) break;
    chunk -= ch;
    kbase += g.in_len[v];
  }
  const int kin = chunk * g.kc + kk;
  int col_base = 0, ov = 0;
  for (; ov < g.n_out - 1; ++ov) {
    const int tv = (g.out_len[ov] + g.ncol - 1) / g.ncol;
    if (nt < tv) break;
    nt -= tv;
    col_base += g.out_len[ov];
  }
  const int cin = nt * g.ncol + cin_local;
  if (kin >= g.in_len[v] || cin >= g.out_len[ov]) return 0.f;
  const int k = kbase + kin, n = col_base + cin;
  if (g.kind != kKindWino) return wsrc_at(w, g.taps, tap, k, n);
  // U = G g G^T: row ra of G down the filter rows, then row rb of G along the filter columns
  const int ra = xi >> 2, rb = xi & 3;
  float t[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float w0 = wsrc_at(w, 9, 0 * 3 + c, k, n), w1 = wsrc_at(w, 9, 1 * 3 + c, k, n), w2 = wsrc_at(w, 9, 2 * 3 + c, k, n);
    t[c] = ra == 0 ? w0 : (ra == 1 ? 0.5f * (w0 + w1 + w2) : (ra == 2 ? 0.5f * (w0 - w1 + w2) : w2));
  }
  return rb == 0 ? t[0] : (rb == 1 ? 0.5f * (t[0] + t[1] + t[2]) : (rb == 2 ? 0.5f * (t[0] - t[1] + t[2]) : t[2]));
}

// slot i of the image: one float, or two bf16 (elements 2i, 2i+1) in the same 4 bytes
__device__ __forceinline__ float image_slot(const PackGeom& g, const unetpp_weight_src& w, long i) {
  if (g.kind != kKindBf16) return image_element(g, w, i);
  return __uint_as_float(pack_bf2(image_element(g, w, 2 * i), image_element(g, w, 2 * i + 1)));
}

__global__ void pack_image_one_kernel(const PackGeom g, const unetpp_weight_src w, float* __restrict__ img) {
  const long i = blockIdx.x * static_cast<long>(blockDim.x) + threadIdx.x;
  if (i < g.floats) img[i] = image_slot(g, w, i);
}

// blockIdx.y = job; every job derives its geometry from its own channel lists
__global__ void pack_image_jobs_kernel(const unetpp_pack_job* __restrict__ jobs) {
  const unetpp_pack_job& j = jobs[blockIdx.y];
  __shared__ PackGeom g;
  if (threadIdx.x == 0) {
    const bool bf = (j.flags & UNETPP_GEMM_BF16) != 0;
    const bool wino = !bf && j.taps == 9 && (j.flags & UNETPP_GEMM_DIRECT) == 0;
    g.kind = bf ? kKindBf16 : (wino ? kKindWino : kKindFast);
    g.taps = j.taps;
    g.kc = bf ? 32 : (wino ? 8 : 16);
    g.ncol = 32;
    g.n_in = j.n_in;
    g.n_out = j.n_out;
    for (int i = 0; i < UNETPP_MAX_VIEWS; ++i) {
      g.in_len[i] = j.in_len[i];
      g.out_len[i] = j.out_len[i];
    }
    finish_geom(g);
  }
  __syncthreads();
  if (g.kind == kKindBf16 && g.taps == 9) {
    __shared__ float stage[9 * 32 * 33];
    const long blocks = g.floats / (9 * 512);
    for (long rc = blockIdx.x; rc < blocks; rc += gridDim.x) fill_bf16_3x3_block(g, j.src, rc, j.image, stage);
    return;
  }
  const long rows = g.floats / (g.kind == kKindWino ? 4096 : 512);
  for (long row = blockIdx.x; row < rows; row += gridDim.x) fill_image_row(g, j.src, row, j.image);
}

bool geom_of(const unetpp_gemm_desc* d, PackGeom& g) {
  FastArgs a;
  const bool bf = d != nullptr && (d->flags & UNETPP_GEMM_BF16) != 0;
  const bool wino = wino_applies(d);
  if (bf ? !bf16_gemm_args(d, a) : !fast_args(d, a, wino ? 8 : 16, 32)) return false;
  g.kind = bf ? kKindBf16 : (wino ? kKindWino : kKindFast);
  g.taps = d->taps;
  g.kc = bf ? 32 : (wino ? 8 : 16);
  g.ncol = 32;
  g.n_in = d->n_in;
  g.n_out = d->n_out;
  for (int i = 0; i < UNETPP_MAX_VIEWS; ++i) {
    g.in_len[i] = i < d->n_in ? d->in[i].c_len : 0;
    g.out_len[i] = i < d->n_out ? d->out[i].c_len : 0;
  }
  finish_geom(g);
  return g.n_chunks == a.n_chunks && g.n_tiles == a.n_tiles;
}

}  // namespace
}  // namespace unetpp

using namespace unetpp;

extern "C" int64_t unetpp_gemm_weight_image_floats(const unetpp_gemm_desc* d) {
  PackGeom g;
  return geom_of(d, g) ? g.floats : 0;
}

extern "C" int unetpp_gemm_pack_weight_image_from(const unetpp_gemm_desc* d, const unetpp_weight_src* src, float* image,
                                                  void* stream) {
  PackGeom g;
  if (src == nullptr || src->src == nullptr || image == nullptr || !geom_of(d, g)) return UNETPP_EINVAL;
  hipLaunchKernelGGL(pack_image_one_kernel, dim3(static_cast<unsigned>((g.floats + 255) / 256)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), g, *src, image);
  return launch_status();
}

extern "C" int unetpp_gemm_pack_weight_image(const unetpp_gemm_desc* d, float* image, void* stream) {
  if (d == nullptr || d->weight == nullptr) return UNETPP_EINVAL;
  long K = 0, N = 0;
  for (int i = 0; i < d->n_in && i < UNETPP_MAX_VIEWS; ++i) K += d->in[i].c_len;
  for (int i = 0; i < d->n_out && i < UNETPP_MAX_VIEWS; ++i) N += d->out[i].c_len;
  unetpp_weight_src w = {};  // the packed [taps][K][Ncols] operand
  w.src = d->weight;
  w.s_t = K * N;
  w.s_k = N;
  w.s_n = 1;
  return unetpp_gemm_pack_weight_image_from(d, &w, image, stream);
}

extern "C" int unetpp_gemm_pack_weight_images(const unetpp_pack_job* jobs_device, int32_t n_jobs, int64_t max_image_floats,
                                              void* stream) {
  if (jobs_device == nullptr || n_jobs < 1 || n_jobs > 65535 || max_image_floats < 1) return UNETPP_EINVAL;
  long bx = (max_image_floats + 256L * 8 - 1) / (256L * 8);  // ~8 elements per thread for the largest image
  if (bx > 256) bx = 256;
  hipLaunchKernelGGL(pack_image_jobs_kernel, dim3(static_cast<unsigned>(bx), static_cast<unsigned>(n_jobs)), dim3(256), 0,
                     static_cast<hipStream_t>(stream), jobs_device);
  return launch_status();
}

/*
 * unetpp_hip.h -- C ABI of the MI355X (gfx950) UNet_Nested forward/backward path.
 *
 * The reference (unanan/UNet_Nested4Tiny_Objects_Keypoints) has no FFI of its own: its hot
 * path is reached through the torch.nn.Module API of models/unet.py and executes inside
 * torch.nn layers.  Each entry point below therefore cites the torch.nn call site in
 * /root/reference/models/unet.py that it replaces.  All functions
 *   - take plain device pointers, explicit sizes and a hipStream_t (passed as void*),
 *   - never allocate, never synchronise, never own memory,
 *   - return 0 on success or a negative UNETPP_E* code (nothing is thrown across the ABI),
 *   - are asynchronous on `stream` and re-entrant per (device, stream).
 * State.  No entry point passes data to another through the library: whatever a launch needs travels in its arguments
 * (the number of BatchNorm rows a convolution wrote reaches its finalize inside unetpp_gemm_fwd by value).  What the
 * library does keep, all of it configuration or diagnostics and none of it per call:
 *   - the CU count of each device (queried once) and, per kernel and device, the "large LDS" opt-in of the runtime;
 *   - unetpp_set_reserved_cus(): one process-wide integer;
 *   - unetpp_debug_set(): a process-wide table of dispatcher switches for A/B runs and tests, initialised ONCE from
 *     UNETPP_* environment variables at the first lookup -- no launch path reads the environment;
 *   - unetpp_last_kernel_name(): a thread-local pointer to a static string (profiling label).
 *
 * Tensor layout on the device is NHWC fp32 ("pixel-major": the channels of one pixel are
 * contiguous).  A `unetpp_view` names a channel slice of such a tensor, optionally sampled
 * on a strided pixel grid; the multi-view GEMM uses that to read a dense-skip concatenation
 * (models/unet.py:198-202) without materialising it, and to express the 2x2/stride-2
 * transposed convolution (models/unet.py:187) as a pointwise GEMM plus pixel shuffle.
 */
#ifndef UNETPP_HIP_H
#define UNETPP_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define UNETPP_ABI_VERSION 11
#define UNETPP_MAX_VIEWS 8

#define UNETPP_OK 0
#define UNETPP_EINVAL (-1)   /* bad argument (null pointer, size <= 0, unsupported shape) */
#define UNETPP_ELAUNCH (-2)  /* hipLaunchKernel reported an error */

/* A channel slice [c_off, c_off + c_len) of an NHWC fp32 tensor [N, Hs, Ws, C], sampled at
 * tensor pixel (y*sy + oy, x*sx + ox) for logical pixel (y, x).  sy = sx = 1, oy = ox = 0
 * is the plain case. */
typedef struct unetpp_view {
  float* ptr;
  int32_t C, c_off, c_len;
  int32_t Hs, Ws;
  int32_t sy, sx, oy, ox;
  /* on load : v = v * scale[c] + shift[c] (c counted inside the slice) when scale != NULL
   * on store: unused */
  const float* scale;
  const float* shift;
  /* ReLU-backward gate: an NHWC tensor with the same geometry as `ptr`; the value is
   * multiplied by (gate > 0).  Applied on load for inputs, on store for outputs. */
  const float* gate;
  int32_t relu;        /* load: max(v, 0) after the affine; store: max(v, 0) after the bias */
  int32_t accumulate;  /* store only: dst += v instead of dst = v */
  int32_t gate_sum;    /* store only, with gate: 0 = gate this contribution before it is added,
                        * 1 = gate the accumulated sum (the last contribution applies the ReLU mask) */
  int32_t reserved;
} unetpp_view;

/* rows = N*H*W logical pixels;  out[p, n] = bias[n] + sum_{tap, k} in[p (+) tap, k] * weight[tap][k][n]
 * K = sum of in[].c_len (virtual channel concatenation in view order), Ncols = sum of out[].c_len.
 * taps = 9: 3x3 window, zero padding 1 (padding is applied AFTER the per-view load transform);
 * taps = 1: pointwise.  Replaces nn.Conv2d(.,.,3,1,1) (models/unet.py:132,140), its input
 * gradient, nn.ConvTranspose2d(.,.,2,2,0) (:187) forward and input gradient, nn.Conv2d(.,.,1) (:191). */
#define UNETPP_GEMM_DIRECT 1 /* flags: direct summation only.  Without it 3x3 launches on the fast path use the
                              * Winograd F(2x2,3x3) form: same fp32 result up to rounding (~1e-7 relative), 2.25x
                              * fewer multiplies.  The flag must be the same when the weight image is packed. */
#define UNETPP_GEMM_BF16 2   /* flags (GEMM and weight-gradient descriptors): bf16 STORAGE -- every activation pointer of
                              * the descriptor (view ptr and gate) addresses bf16 NHWC data although it is typed
                              * float*; C, c_off, c_len count bf16 channels and must be multiples of 8.  bias, scale,
                              * shift, stats_partial and slabs stay fp32; the arithmetic is v_mfma_f32_32x32x16_bf16
                              * with fp32 accumulation (BASELINE configs[3]/[4]).  There is no generic bf16 kernel:
                              * a descriptor the MFMA kernels cannot take returns UNETPP_EINVAL. */
/* BatchNorm2d training-mode finalize (models/unet.py:133) attached to the convolution call that takes the statistics:
 * the call leaves mean, invstd, scale = gamma * invstd, shift = beta - mean * scale ([Ncols] each) and the updated
 * running statistics behind, exactly as a unetpp_bn_finalize call of the caller would.  The persistent kernels then
 * write ONE row of sums per workgroup (all its units; <= 2048 rows instead of one per 256-pixel block) and the library
 * enqueues the finalize over those rows itself; kernels without that epilogue (generic shapes, more than 256
 * columns) keep per-block rows.  Same result up to the row partition of the fp32 sums.
 * scale == NULL: not attached (the caller finishes stats_partial with unetpp_bn_finalize). */
typedef struct unetpp_bn_fused {
  const float* gamma;
  const float* beta;
  float* running_mean; /* both NULL or both set */
  float* running_var;
  float* mean;
  float* invstd;
  float* scale;
  float* shift;
  int64_t count;     /* N*H*W */
  float eps, momentum;
} unetpp_bn_fused;

typedef struct unetpp_gemm_desc {
  int32_t N, H, W;
  int32_t taps;
  int32_t n_in;
  int32_t n_out;
  int32_t flags;    /* UNETPP_GEMM_* */
  int32_t reserved; /* 0 */
  unetpp_view in[UNETPP_MAX_VIEWS];
  unetpp_view out[UNETPP_MAX_VIEWS];
  const float* weight; /* packed [taps][K][Ncols] (see unetpp_pack_weight); may be NULL when weight_image is set */
  const float* bias;   /* [Ncols] or NULL */
  /* optional BatchNorm statistics epilogue: per pixel-block partial (sum, sum of squares) of the
   * stored values, [unetpp_gemm_pixel_blocks()][Ncols][2]; requires n_out == 1.  With `bn` set the buffer is
   * workspace of [unetpp_gemm_stats_rows()][Ncols][2] floats whose row structure is the kernel's own. */
  float* stats_partial;
  /* optional fast path: `weight` re-laid as the kernel's LDS image by unetpp_gemm_pack_weight_image
   * (unetpp_gemm_weight_image_floats() floats).  NULL selects the generic kernel. */
  const float* weight_image;
  unetpp_bn_fused bn; /* bn.scale != NULL: finish the statistics inside this call (needs stats_partial) */
} unetpp_gemm_desc;

/* weight gradient:  dW[tap][k][n] = sum_p x[p (+) tap, k] * dy[p, n]  (+ db[n] = sum_p dy[p, n]).
 * Replaces the weight/bias gradient of nn.Conv2d 3x3 / 1x1 and nn.ConvTranspose2d 2x2 s2
 * (autograd of models/unet.py:132,140,187,191).  Split over pixel blocks: each of the
 * `n_split` blocks per (k-tile, n-tile) writes one partial slab; unetpp_wgrad_finish sums them. */
typedef struct unetpp_wgrad_desc {
  int32_t N, H, W;
  int32_t taps;
  int32_t n_x;
  int32_t n_dy;
  unetpp_view x[UNETPP_MAX_VIEWS];  /* K = sum c_len */
  unetpp_view dy[UNETPP_MAX_VIEWS]; /* Ncols = sum c_len (several views: the 4 pixel phases of a 2x2 deconv) */
  int32_t n_split;                  /* partial slabs (<= unetpp_wgrad_max_split) */
  int32_t flags;                    /* UNETPP_GEMM_DIRECT: direct summation only (no Winograd) */
  float* slabs;                    /* [n_split][planes*K + 1][Ncols], planes = unetpp_wgrad_slab_planes(); last row = db */
} unetpp_wgrad_desc;

int unetpp_abi_version(void);
const char* unetpp_build_arch(void); /* "gfx950" */

/* Name of the device kernel the calling thread's most recent unetpp_gemm_fwd / unetpp_wgrad call launched
 * (profiling labels; equals the rocprofv3 kernel name up to template arguments).  Static string, "" before the
 * first call.  Diagnostic only (thread-local). */
const char* unetpp_last_kernel_name(void);

/* Dispatcher switches for A/B measurements and for tests that hold two kernels against each other inside one process
 * (v9; replaces per-launch getenv).  `name` is the switch without its UNETPP_ prefix: BF16_NO_DMA, BF16_DMA_ALL,
 * BF16_DMA_MIN8, BF16_DMA_FORM, BF16_DMA_SMALL, BF16_DMA_STATS, BF16_DMA_POINTWISE, BF16_DMA_SPLIT, BF16_WGRAD_QUAD,
 * WINO_NO_LEAN, WINO_ONE_PER_CU, MEMSET_NODES, BF16_PW_PLAIN, PW_DIRECT (0: fp32 pointwise launches back on the LDS-staged
 * kernels), PW_NT (fp32 pointwise GEMM: 1 / 0 = non-temporal / plain stores whatever the output size),
 * HEAD_WGS_PER_CU (bf16 head backward: at most this many workgroups per CU take tiles, default 4; 0 = the whole grid).  set != 0: the switch takes `value`; set == 0: back to the
 * dispatcher's built-in default.  The environment variable UNETPP_<name>, if present when the library first looks a
 * switch up, is the initial setting.  Process-wide; results never depend on a switch beyond the summation order of the
 * kernel it selects.  UNETPP_EINVAL for an unknown name. */
int unetpp_debug_set(const char* name, int64_t value, int32_t set);
/* The current state of a switch (v10): returns 1 and stores its value in *value when the switch is set (by
 * unetpp_debug_set or by its environment variable), 0 when it is unset (*value untouched), UNETPP_EINVAL for an unknown
 * name.  Lets a scoped override put back what it found. */
int unetpp_debug_get(const char* name, int64_t* value);

/* Data-parallel co-scheduling knob (v8).  Every hot kernel is a persistent launch sized to fill all CUs at 2-3 workgroups
 * per CU, so a collective kernel (RCCL all-reduce of a gradient bucket on a side stream, trainer/trainer.py:338's
 * replicas replaced by one process per GPU) only gets waves when a compute kernel ends.  n > 0 makes the persistent
 * grids leave n CUs' worth of workgroups unlaunched (a caller of unetpp_wgrad sizes n_split itself and should aim at
 * CUs - n workgroups, as the Python layer does); 0 (default,
 * or the UNETPP_RESERVED_CUS environment variable read at the first call) = use every CU.  Process-wide, takes effect
 * at the next launch; results do not depend on it.  Returns the value now in force (n clamped to [0, CUs - 8]);
 * unetpp_set_reserved_cus(-1) only reads it. */
int32_t unetpp_set_reserved_cus(int32_t n);
/* CUs of the current device a persistent grid may fill (physical - reserved, at least 8); *physical (may be NULL) gets
 * the device's CU count.  A caller that sizes unetpp_wgrad's n_split scales its target by usable / physical (v9). */
int32_t unetpp_usable_cus(int32_t* physical);

/* ---- multi-view pixel GEMM on MFMA (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32) ------ */
int64_t unetpp_gemm_pixel_blocks(int32_t N, int32_t H, int32_t W);
/* rows of [Ncols][2] floats stats_partial must hold when the finalize is fused (d->bn.scale != NULL) */
int64_t unetpp_gemm_stats_rows(int32_t N, int32_t H, int32_t W);
int unetpp_gemm_fwd(const unetpp_gemm_desc* d, void* stream);
/* Fast path (register-prefetched LDS-image kernels; Winograd F(2x2,3x3) for taps = 9 unless UNETPP_GEMM_DIRECT):
 * applies when every input view is a 16-byte aligned slice without a ReLU gate on load (C, c_off and c_len
 * multiples of 4; an affine + ReLU load transform is allowed).  Returns the image size in floats, or 0 when the
 * descriptor must use the generic kernel. */
int64_t unetpp_gemm_weight_image_floats(const unetpp_gemm_desc* d);
int unetpp_gemm_pack_weight_image(const unetpp_gemm_desc* d, float* image, void* stream); /* from d->weight (packed) */
/* The same image built straight from a parameter in its torch layout (no unetpp_pack_weight pass; d->weight may be
 * NULL for a launch that carries a weight image).  Element W[tap][k][n] of the GEMM weight is read at
 *   src[tap' * s_t + (k % k_inner) * s_k + (k / k_inner) * s_ko + (n % n_inner) * s_n + (n / n_inner) * s_no],
 * tap' = flip ? taps-1-tap : tap; k_inner / n_inner = 0 mean "no split".  nn.Conv2d [co,ci,3,3] forward:
 * (s_t,s_k,s_n) = (1, 9, 9ci); its input gradient: flip, (1, 9ci, 9); nn.ConvTranspose2d [ci,co,2,2] forward
 * (N = phase*co + c): s_k = 4co, n_inner = co, s_n = 4, s_no = 1; its input gradient (K = phase*co + c): k_inner = co,
 * s_k = 4, s_ko = 1, s_n = 4co. */
typedef struct unetpp_weight_src {
  const float* src;
  int64_t s_t, s_k, s_ko, s_n, s_no;
  int32_t k_inner, n_inner, flip, reserved;
} unetpp_weight_src;
int unetpp_gemm_pack_weight_image_from(const unetpp_gemm_desc* d, const unetpp_weight_src* src, float* image, void* stream);
/* Many images in ONE launch (all launches of a forward or backward pass): a table of jobs in DEVICE memory.  A job
 * carries what the image layout depends on -- taps, flags and the channel counts of the launch's input / output views
 * (must describe a launch for which unetpp_gemm_weight_image_floats() > 0) -- plus source and destination.
 * max_image_floats = the largest image of the table (sizes the grid). */
typedef struct unetpp_pack_job {
  unetpp_weight_src src;
  float* image;
  int32_t taps, flags, n_in, n_out;
  int32_t in_len[UNETPP_MAX_VIEWS], out_len[UNETPP_MAX_VIEWS];
} unetpp_pack_job;
int unetpp_gemm_pack_weight_images(const unetpp_pack_job* jobs_device, int32_t n_jobs, int64_t max_image_floats,
                                   void* stream);

int32_t unetpp_wgrad_max_split(int32_t N, int32_t H, int32_t W);
/* Planes per slab the kernel chosen for this descriptor writes: `taps` for direct summation, 16 for the Winograd
 * F(2x2,3x3) kernel (3x3, plain 16-byte aligned views, images at least 17 wide, no UNETPP_GEMM_DIRECT in flags): it
 * accumulates transform-domain products and unetpp_wgrad_finish(taps = 16) maps them back to the 9 taps. */
int32_t unetpp_wgrad_slab_planes(const unetpp_wgrad_desc* d);
/* (32-channel, 32-column) tile pairs one workgroup of the kernel chosen for this descriptor owns: 1, or 4 for the bf16
 * kernel of layers whose views are all multiples of 64 channels wide (one workgroup per CU).  unetpp_wgrad launches
 * n_split * pairs / this workgroups; the caller sizes n_split with it. */
int32_t unetpp_wgrad_pairs_per_workgroup(const unetpp_wgrad_desc* d);
int unetpp_wgrad(const unetpp_wgrad_desc* d, void* stream);
/* `taps` = planes per slab.  column n = o*n_inner + i:  dw[t*d_t + k*d_k + i*d_n + o*d_o] = sum_s slabs[s][t*K + k][n];
 * db[i] = sum_o sum_s slabs[s][taps*K][o*n_inner + i]   (n_inner = Ncols for a plain convolution) */
int unetpp_wgrad_finish(const float* slabs, int32_t n_split, int32_t taps, int32_t K, int32_t Ncols, int32_t n_inner,
                        float* dw, int64_t d_t, int64_t d_k, int64_t d_n, int64_t d_o, float* db, void* stream);

/* dst[t*d_t + k*d_k + n*d_n] = src[tt*s_t + k*s_k + n*s_n], tt = flip ? T-1-t : t.
 * Re-lays a torch-layout weight ([co,ci,3,3] / [ci,co,2,2], the state-dict contract of
 * models/unet.py) into the packed [taps][K][Ncols] operand of the GEMM. */
int unetpp_pack_weight(float* dst, const float* src, int32_t T, int32_t K, int32_t Ncols,
                       int64_t d_t, int64_t d_k, int64_t d_n, int64_t s_t, int64_t s_k, int64_t s_n,
                       int32_t flip, void* stream);
/* Many strided copies in ONE launch: dst[o*dst_stride + i] = src[o*src_stride + i], o < n_outer, i < n_inner (floats;
 * n_outer*n_inner < 2^31 per job; 16-byte pieces when pointers, n_inner and strides allow).  The job table lives in
 * DEVICE memory; max_elems = the largest n_outer*n_inner of the table (sizes the grid).  Host-side re-layouts of a pass
 * that are not weight images: the bias of a 2x2 transposed convolution repeated for its four pixel phases (the GEMM
 * column is phase*co + c: src_stride 0, n_outer 4), the input-gradient weights of the dense skips grouped by producer
 * (channel slices of the consumers' conv1 weights, the torch.cat of models/unet.py:198-202, stacked along the output-channel axis). */
typedef struct unetpp_copy_job {
  const float* src;
  float* dst;
  int64_t n_outer, n_inner, src_stride, dst_stride;
} unetpp_copy_job;
int unetpp_copy_jobs(const unetpp_copy_job* jobs_device, int32_t n_jobs, int64_t max_elems, void* stream);

/* ---- BatchNorm2d, training mode (models/unet.py:133) -------------------------------------- */
/* partial [n_blocks][C][2] (sum, sumsq) -> mean, invstd, scale = gamma*invstd, shift = beta - mean*scale;
 * running_mean/var updated with `momentum` (unbiased variance), all [C]. count = N*H*W. */
int unetpp_bn_finalize(const float* partial, int64_t n_blocks, int32_t C, int64_t count,
                       const float* gamma, const float* beta, float eps, float momentum,
                       float* running_mean, float* running_var,
                       float* mean, float* invstd, float* scale, float* shift, void* stream);
/* eval mode: scale/shift from running statistics */
int unetpp_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                          const float* running_var, float eps, int32_t C, float* scale, float* shift,
                          void* stream);
/* act = relu?(y*scale + shift) (scale may be NULL = identity);  optional fused nn.MaxPool2d(2)
 * (models/unet.py:219): pooled [N,H/2,W/2,C] and pool_idx (uint8 window index 0..3, first max wins).
 * act may be NULL when only the pooled output is wanted; pool_idx may be NULL when the winners are not (forward-only
 * callers; backward needs them).  Odd H or W: the window grid is floor(H/2) x floor(W/2) as in
 * nn.MaxPool2d (the last row / column is in no window; act still covers every pixel). */
int unetpp_affine_relu_pool(const float* y, const float* scale, const float* shift, int32_t relu,
                            int32_t N, int32_t H, int32_t W, int32_t C,
                            float* act, float* pooled, uint8_t* pool_idx, void* stream);
/* d_act[window argmax] += d_pooled */
int unetpp_maxpool_bwd(const float* d_pooled, const uint8_t* pool_idx, int32_t N, int32_t H, int32_t W,
                       int32_t C, float* d_act, void* stream);
int64_t unetpp_bn_bwd_blocks(int64_t pixels, int32_t C);
/* g = d_act * (y*scale+shift > 0); partial [blocks][C][2] = (sum g, sum g*xhat) */
int unetpp_bn_bwd_reduce(const float* d_act, const float* y, const float* scale, const float* shift,
                         const float* mean, const float* invstd, int64_t pixels, int32_t C,
                         float* partial, void* stream);
/* sums the partials -> dgamma, dbeta; then dy = gamma*invstd*(g - dbeta/M - xhat*dgamma/M) (dy may alias d_act) */
int unetpp_bn_bwd_finalize(const float* partial, int64_t n_blocks, int32_t C, float* dgamma, float* dbeta,
                           void* stream);
int unetpp_bn_bwd_apply(const float* d_act, const float* y, const float* scale, const float* shift,
                        const float* mean, const float* invstd, const float* gamma,
                        const float* dgamma, const float* dbeta, int64_t pixels, int32_t C,
                        float* dy, void* stream);

/* BatchNorm backward of an encoder node whose output was also 2x2 max-pooled (models/unet.py:257-263): the pooled
 * consumer's gradient d_pooled [N, H/2, W/2, C] is routed to the argmax recorded by unetpp_affine_relu_pool WHILE
 * d_act is read, in both passes, instead of a separate unetpp_maxpool_bwd pass over d_act.  Needs
 * unetpp_bn_bwd_pool_ok(N, H, W, C) (4-aligned C with a power-of-two C/4 <= 256, even H and W, < 2^31 elements);
 * otherwise call unetpp_maxpool_bwd and the plain functions.  partial: unetpp_bn_bwd_blocks(N*H*W, C) rows, finished
 * by unetpp_bn_bwd_finalize as usual. */
int unetpp_bn_bwd_pool_ok(int32_t N, int32_t H, int32_t W, int32_t C);
int unetpp_bn_bwd_reduce_pool(const float* d_act, const float* y, const float* scale, const float* shift,
                              const float* mean, const float* invstd, const float* d_pooled, const uint8_t* pool_idx,
                              int32_t N, int32_t H, int32_t W, int32_t C, float* partial, void* stream);
int unetpp_bn_bwd_apply_pool(const float* d_act, const float* y, const float* scale, const float* shift,
                             const float* mean, const float* invstd, const float* gamma, const float* dgamma,
                             const float* dbeta, const float* d_pooled, const uint8_t* pool_idx, int32_t N, int32_t H,
                             int32_t W, int32_t C, float* dy, void* stream);

/* ---- deep-supervision head: sigmoid(Conv1x1(Dropout(x))) (models/unet.py:242-244,254,283-286) ---- */
/* x NHWC [P, C]; weight [n_cls, C]; out NCHW [N, n_cls, H, W].  Dropout: keep mask regenerated from
 * (seed, element index) with keep probability 1-p_drop, or read from `mask` (uint8 NHWC) when not NULL;
 * p_drop = 0 disables it.  seed_dev (v8, may be NULL): one uint64 in device memory that is ADDED to `seed` when the
 * kernel runs -- a launch captured in a HIP graph bakes `seed` in, so a replayed training step keeps the varying part
 * of its seed there (forward and backward of a step must see the same value). */
int unetpp_head_fwd(const float* x, const float* weight, const float* bias, int32_t N, int32_t H, int32_t W,
                    int32_t C, int32_t n_cls, float p_drop, uint64_t seed, const uint8_t* mask, const uint64_t* seed_dev,
                    float* out_nchw, void* stream);
int64_t unetpp_head_bwd_blocks(int64_t pixels);
/* d_out, out: NCHW.  dx (NHWC) is written (accumulate = 0) or added to; partial [blocks][n_cls*C + n_cls]. */
int unetpp_head_bwd(const float* d_out_nchw, const float* out_nchw, const float* x, const float* weight,
                    int32_t N, int32_t H, int32_t W, int32_t C, int32_t n_cls, float p_drop, uint64_t seed,
                    const uint8_t* mask, const uint64_t* seed_dev, float* dx, int32_t accumulate, int32_t gate_x, float* partial,
                    void* stream); /* gate_x: after the (optional) accumulate, dx *= (x > 0) -- the ReLU mask of x */
/* out[i] = sum_b partial[b][i], i < len (used for head dW/db) */
int unetpp_sum_partials(const float* partial, int64_t n_blocks, int64_t len, float* out, void* stream);

/* ---- bilinear x2, align_corners=True (nn.UpsamplingBilinear2d, models/unet.py:190) ---------- */
int unetpp_bilinear2x_fwd(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, float* y, void* stream);
int unetpp_bilinear2x_bwd(const float* dy, int32_t N, int32_t H, int32_t W, int32_t C, float* dx, int32_t accumulate,
                          void* stream);

/* ---- caller side of the training step (trainer/trainer.py:114-136) -------------------------- */
/* FocalLoss_BCE_2d (tools/losses/focal_loss.py:255-301, size_average = False), value and gradient in one pass:
 * e = 1 - |pred - target| + 1e-20;  loss = sum(-(1 - e)^gamma log e) / rows;  rows = N*C of the [N,C,H,W] heads.
 * partial[unetpp_focal_bce_blocks(n)] is workspace (per-block sums, added in fixed order); loss[0] receives the
 * value, grad (may be NULL) dloss/dpred.  pred/target/grad 16-byte aligned, n elements. */
int64_t unetpp_focal_bce_blocks(int64_t n);
int unetpp_focal_bce(const float* pred, const float* target, int64_t n, int64_t rows, float gamma, float* grad,
                     float* partial, float* loss, void* stream);
/* The trainer's loop over the deep-supervision heads (trainer/trainer.py:122-135: criterion on every head, mean over
 * heads) in one pass over the target: loss[1 + h] = FocalLoss_BCE_2d(pred[h], target) exactly as unetpp_focal_bce
 * computes it, loss[0] = (0 + loss[1] + ... + loss[n_heads]) * (1 / n_heads) in that order (float32, what the loop body's
 * tensor arithmetic does), grad[h] = d loss[0] / d pred[h] = (d loss[1 + h] / d pred[h]) * (1 / n_heads) -- the product
 * autograd forms from the same two float32 factors.  1 <= n_heads <= UNETPP_MAX_HEADS; partial[n_heads *
 * unetpp_focal_bce_blocks(n)] is workspace; every pred[h] / grad[h] (grad[h] may be NULL) 16-byte aligned, n elements. */
#define UNETPP_MAX_HEADS 8
typedef struct unetpp_focal_heads {
  const float* pred[UNETPP_MAX_HEADS];
  float* grad[UNETPP_MAX_HEADS];
  int32_t n_heads, reserved;
} unetpp_focal_heads;
int unetpp_focal_bce_heads(const unetpp_focal_heads* heads, const float* target, int64_t n, int64_t rows, float gamma,
                           float* partial, float* loss, void* stream);
/* create_heatmap (tools/misc/helper.py:87-172): key points [N][P][2] as (x, y), P >= 6 -> float32 [N,4,H,W]:
 * ch0 = point 0, ch1 = points 1..3 summed / max, ch2 = point 4, ch3 = points 5..P-1 summed / max, each map
 * exp(-0.5 sqrt(dx^2 + dy^2) / radius) (the reference uses radius 3).  workspace: unetpp_heatmap_workspace_bytes(). */
int64_t unetpp_heatmap_workspace_bytes(int32_t N, int32_t H, int32_t W);
int unetpp_create_heatmap(const float* points, int32_t N, int32_t P, int32_t H, int32_t W, float radius,
                          float* out_nchw, void* workspace, void* stream);

/* ---- layout converters at the network edge ------------------------------------------------ */
int unetpp_nchw_to_nhwc(const float* src, int32_t N, int32_t C, int32_t H, int32_t W, float* dst, void* stream);
int unetpp_nhwc_to_nchw(const float* src, int32_t N, int32_t C, int32_t H, int32_t W, float* dst, void* stream);

/* ---- bf16-storage companions of the UNETPP_GEMM_BF16 launches (BASELINE configs[3]/[4]) -------------
 * Same operations as their fp32 namesakes above, on bf16 NHWC activations (void*: 16-byte aligned, C a multiple of
 * 8; the BatchNorm backward and head kernels want C/8 a power of two); coefficients, partial sums, parameter
 * gradients and the heads' NCHW probabilities / their gradients stay fp32.  The max-pool winner is the first maximum
 * in scan order of the STORED (bf16-rounded) activation.  d_pooled / pool_idx may be NULL (no pooled consumer):
 * otherwise the pooled gradient is routed to the window argmax while d_act is read, in both passes. */
int unetpp_affine_relu_pool_bf16(const void* y, const float* scale, const float* shift, int32_t relu,
                                 int32_t N, int32_t H, int32_t W, int32_t C,
                                 void* act, void* pooled, uint8_t* pool_idx, void* stream);
int64_t unetpp_bn_bwd_blocks_bf16(int64_t pixels, int32_t C);
int unetpp_bn_bwd_reduce_bf16(const void* d_act, const void* y, const float* scale, const float* shift,
                              const float* mean, const float* invstd, const void* d_pooled, const uint8_t* pool_idx,
                              int32_t N, int32_t H, int32_t W, int32_t C, float* partial, void* stream);
int unetpp_bn_bwd_apply_bf16(const void* d_act, const void* y, const float* scale, const float* shift,
                             const float* mean, const float* invstd, const float* gamma, const float* dgamma,
                             const float* dbeta, const void* d_pooled, const uint8_t* pool_idx,
                             int32_t N, int32_t H, int32_t W, int32_t C, void* dy, void* stream);
int unetpp_head_fwd_bf16(const void* x, const float* weight, const float* bias, int32_t N, int32_t H, int32_t W,
                         int32_t C, int32_t n_cls, float p_drop, uint64_t seed, const uint8_t* mask, const uint64_t* seed_dev,
                         float* out_nchw, void* stream);
/* partial: unetpp_head_bwd_blocks(N*H*W) rows of [n_cls*C + n_cls], finished by unetpp_sum_partials */
int unetpp_head_bwd_bf16(const float* d_out_nchw, const float* out_nchw, const void* x, const float* weight,
                         int32_t N, int32_t H, int32_t W, int32_t C, int32_t n_cls, float p_drop, uint64_t seed,
                         const uint8_t* mask, const uint64_t* seed_dev, void* dx, int32_t accumulate, int32_t gate_x, float* partial,
                         void* stream);

/* is_batchnorm=False in bf16: d_act += d_pooled at the window argmax (unetpp_affine_relu_pool_bf16's pool_idx), then,
 * with gate != NULL, d_act *= (gate > 0) -- the ReLU mask of the node, applied by its last gradient contribution */
int unetpp_maxpool_bwd_bf16(const void* d_pooled, const uint8_t* pool_idx, int32_t N, int32_t H, int32_t W, int32_t C,
                            void* d_act, const void* gate, void* stream);
/* is_deconv=False in bf16 (nn.UpsamplingBilinear2d, models/unet.py:190): x [N,H,W,C] -> y [N,2H,2W,C]; backward in
 * gather form: dx = (accumulate ? dx : 0) + stencil^T(dy), then dx *= (gate > 0) when gate != NULL */
int unetpp_bilinear2x_fwd_bf16(const void* x, int32_t N, int32_t H, int32_t W, int32_t C, void* y, void* stream);
int unetpp_bilinear2x_bwd_bf16(const void* dy, int32_t N, int32_t H, int32_t W, int32_t C, void* dx, int32_t accumulate,
                               const void* gate, void* stream);

/* Input gradient of the network's first convolution (models/unet.py:220, 1..4 input channels) with bf16 activation
 * storage: dy bf16 [N,H,W,COUT] (COUT a multiple of 8, <= 128), weight fp32 in its torch layout [COUT][CIN][3][3],
 * dx fp32 [N,H,W,CIN].  (fp32 storage: unetpp_gemm_fwd with the rotated weights, as for every other layer.) */
int unetpp_first_layer_dgrad_bf16(const void* dy, const float* weight, int32_t N, int32_t H, int32_t W, int32_t CIN,
                                  int32_t COUT, float* dx, void* stream);

/* ---- heat-map side of validation (tools/misc/heatmap.py; SURVEY 8 row f3 -- parity unpinned: the reference needs
 * OpenCV, absent from the build image) ------------------------------------------------------------------------------
 * unetpp_heatmap_pattern: Heatmap.create_heatmap (heatmap.py:203-230).  points [N, P, 2] as (x, y); map m draws the
 * key points map_points[map_begin[m] .. map_begin[m+1]) (device int32 arrays, n_maps + 1 begins); out [N, n_maps, H, W]:
 * float32 sum of exp(-0.5 * distance / radius) in pattern order, divided by the map's maximum. */
int64_t unetpp_heatmap_pattern_workspace_bytes(int32_t N, int32_t n_maps, int32_t H, int32_t W);
int unetpp_heatmap_pattern(const float* points, int32_t N, int32_t P, const int32_t* map_points,
                           const int32_t* map_begin, int32_t n_maps, int32_t H, int32_t W, float radius,
                           float* out_nchw, void* workspace, void* stream);
/* unetpp_keypoints_extract: Heatmap.extract_points_ (heatmap.py:148-200).  heat [maps, H, W]; thr [maps] (device).
 * Stages on one workspace: 0 = mask (values < thr zeroed, 3x3 median > 0) and label initialisation; 1 = `sweeps` rounds
 * of label merging (8-connected components), *changed (device int32, zeroed by the caller) is set while labels still
 * move: repeat until it stays 0; 2 = per-region maximum and selection: points [maps, num, 2] = (x, y) of the first pixel
 * in raster order attaining the maximum of the num brightest regions (ties: raster order of the regions' first
 * pixels), -1 where there are fewer; counts [maps] = regions found (every count is ranked exactly; max_regions only
 * sizes the fast path's buffer).  0, 1.., 2 alone take a region to be a component of the mask.  The reference's own
 * region step (region_segment_, heatmap.py:100-144: distance-transform cores grown back by a watershed, so that blobs
 * which touch are split) goes between: 0; 3 = 3x3 chamfer distance initialised; 4 = `sweeps` relaxation sweeps, repeat
 * until *changed stays 0; 5 = cores (distance > 0.1 * the map's maximum) as labels; 1.. = components of the cores;
 * 6 = markers (core root / background beyond the two-pixel ring / unknown); 7 = `sweeps` pairs of flood rounds, repeat
 * until *changed stays 0; 8 = region labels; 2.
 * Workspace layout (for callers that want the regions themselves, region_segment_'s return value): uint64 best
 * [maps*H*W], uint64 [maps*max_regions*2], then int32 label [maps*H*W] (after stage 8 / the last stage 1: raster index
 * of the region's first pixel, -1 outside), int32 distance [maps*H*W] (16-bit fixed point), int32 marker [maps*H*W]. */
int64_t unetpp_keypoints_workspace_bytes(int32_t maps, int32_t H, int32_t W, int32_t max_regions);
int unetpp_keypoints_extract(int32_t stage, const float* heat, int32_t maps, int32_t H, int32_t W,
                             const float* thr_per_map, int32_t num, int32_t max_regions, int32_t sweeps,
                             void* workspace, int32_t* changed, float* points, int32_t* counts, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* UNETPP_HIP_H */

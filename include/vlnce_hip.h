/*
 * vlnce_hip.h -- C ABI of libvlnce_hip.so: hand-written CDNA4 (gfx950) kernels
 * for VLN-CE's per-step policy hot path (vlnce_baselines/models/ of the
 * reference).  The reference is pure Python on torch/cuDNN and has no FFI of
 * its own; each entry point below names the torch operator call site in the
 * reference that it replaces (file:line relative to /root/reference).
 *
 * Conventions
 *   - every function returns 0 on success, non-zero on error; the message is
 *     available through vlnce_last_error() (thread-local).
 *   - all pointers are DEVICE pointers valid on `stream`; the library never
 *     allocates, never synchronises and never reads the environment.  Its only
 *     mutable state is the thread-local error string and the process-wide
 *     dispatch options below (vlnce_set_option), which select between kernels
 *     that compute the same function.  Workspaces are caller-allocated.
 *   - activations are fp32, channels-last: images [N,H,W,C], matrices row-major.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream).
 */
#ifndef VLNCE_HIP_H
#define VLNCE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vlnce_stream_t;

enum { VLNCE_ACT_NONE = 0, VLNCE_ACT_RELU = 1, VLNCE_ACT_SIGMOID = 2, VLNCE_ACT_TANH = 3 };

int vlnce_version(void); /* major*100 + minor; 143 = this header */
int vlnce_option_count(void);              /* length of vlnce_prologue.options                        */
int vlnce_option_index(const char* name);  /* index of a named dispatch option in it, -1 if unknown   */
const char* vlnce_last_error(void);

/* Dispatch options: which of the library's equivalent kernels a launch is given to.  Explicit
 * state instead of environment variables, so that a host (or a test) can force a kernel onto a
 * shape in-process and restore the default afterwards.  No reference counterpart (torch picks its
 * cuDNN / MIOpen algorithm through torch.backends.cudnn.benchmark, never set by the reference).
 *   name               default  meaning
 *   "conv_math"        2        0: v_mfma_f32_32x32x2_f32 only; non-zero: the plane kernels (conv_p3 /
 *                               conv_u3 / conv_s3 / conv_m3 / conv_x3 / stem7) on the 16-bit matrix pipe.
 *                               The library tests zero / non-zero only; the HOST reads the value as the
 *                               plane format in which it packs the weights (vlnce_prologue.w_format):
 *                               1 = three bf16 planes, six products per multiply (fp32's exponent range);
 *                               2 = fp16 planes, THREE products per multiply (ABI 142; |x| < 65504,
 *                               |w| < 32, forward operands -- see vlnce_conv2d_pack_weights)
 *   "p3"               2        conv_p3_kernel: 0 off, 1 every layer it covers, 2 KxK only, 3 1x1 only
 *   "p3_tile"          0        0: by CU fill; 1..6: forced tile shape
 *   "s3"               1        conv_s3_kernel (short-K wide 1x1): 0 off, 1 default rule, 2 every eligible shape
 *   "u3"               1        conv_u3_kernel (wide 1x1): 0 off, 1 default rule, 2 / 3 force 64- / 128-row tiles
 *   "u3_waves"         8        8, or 4 (one wave per SIMD)
 *   "x3_tile"          0        conv_x3_kernel: 0 by CU fill, 1..4 forced tile shape
 *   "igemm_tile"       0        igemm_kernel: 0 rule, 1 128x128, 2 128x64, 3 64x64
 *   "igemm_nobuf"      0        1: no buffer-descriptor operand loads
 *   "igemm_no_splitk"  0        1: no split-K
 *   "wgrad_tile"       64       vlnce_conv2d_wgrad: 1 = the fp32-MFMA kernel everywhere (64x64 tiles; A/B against the
 *                               bf16-plane kernel that takes Cin % 32 == 0, Cout % 32 == 0 layers otherwise), 64 / 128 =
 *                               tile of the fp32-MFMA kernel where that one runs
 *   "rollout_one_xcd"  0        1: all workgroups of vlnce_gru_rollout_* on one XCD
 *   "m3"               1        conv_m3_kernel (small launches): 0 off, 1 default rule, 2 / 3 every layer it covers
 * Set options between launches, not concurrently with them (relaxed atomics).  Unknown names
 * return non-zero. */
int vlnce_set_option(const char* name, int value);
int vlnce_get_option(const char* name, int* value);
int vlnce_option_default(const char* name, int* value);

/* ---------------------------------------------------------------- conv / GEMM
 * Implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 (exact fp32):
 *   Y[m, co] = epilogue( sum_{r,q,ci} prologue(X[n, ho*s-p+r, wo*s-p+q, ci]) * W[co, r, q, ci] )
 * with m = (n*Ho + ho)*Wo + wo.  Replaces nn.Conv2d(+BatchNorm2d/+ReLU/+add) of
 * the torchvision trunk (models/encoders/resnet_encoders.py:136-139,199) and of
 * habitat's GroupNorm ResNetEncoder (resnet_encoders.py:31-43,95).
 */
typedef struct {
  int N, H, W, Cin;   /* input  [N,H,W,Cin], pixel stride ldx floats (>= Cin) */
  int Cout, KH, KW;   /* weight [Cout,KH,KW,Cin] ("OHWI"), contiguous          */
  int stride, pad;
  int Ho, Wo;         /* output [N,Ho,Wo,Cout], row stride ldy floats (>= Cout) */
  int ldx, ldy;
} vlnce_conv_desc;

typedef struct {
  /* input transform applied in the A-operand loader BEFORE zero padding:
   *   x' = act((x - in_center[ci]) * in_scale[ci] + in_shift[ci]);  NULL scale = identity,
   *   NULL center = 0.
   * Used for the RGB stem's /255 (+ImageNet mean/std, resnet_encoders.py:171-192)
   * and to apply the previous layer's BatchNorm+ReLU on the fly. */
  const float* in_scale;
  const float* in_shift;
  const float* in_center;
  int in_relu;
  /* Optional second input (1x1 / stride-1 / pad-0 convolutions with Cin % 32 == 0 only): the
   * end of a residual block, out = relu(bn3(conv3) + identity) (torchvision Bottleneck.forward),
   * is evaluated inside the NEXT block's first convolution instead of in a pass of its own:
   *   x' = act((x - in_center)*in_scale + in_shift + ((x2 - in2_center)*in2_scale + in2_shift))
   * x2 has the layout of x; in2_scale NULL = x2 is added as is (identity skip), non-NULL = the
   * downsample branch's raw conv output with its BatchNorm.  side_out (same layout as x, or
   * NULL) receives x' -- the block output the following layers read as their skip input. */
  const float* x2;
  const float* in2_scale;
  const float* in2_shift;
  const float* in2_center;
  float* side_out;
  /* Optional: the weights w_ohwi pre-split into three bf16 planes by
   * vlnce_conv2d_split_weights() (layout [3][Cout*KH*KW*Cin] of 16-bit words).  With it,
   * convolutions on the buffer-descriptor hot path (Cin % 32 == 0) run as six bf16 plane
   * products on the bf16 matrix pipe (conv_x3_kernel: fp32-equivalent result, DESIGN.md
   * section 6); without it (NULL) they run on v_mfma_f32_32x32x2_f32. */
  const void* w_split;
  /* Optional: the weights as bf16-plane MFMA B fragments (vlnce_conv2d_pack_weights()).  With it,
   * stride-1 KxK and all 1x1 convolutions with Cin % 32 == 0 and Cout % 32 == 0 run on
   * conv_p3_kernel (A operand transformed once per workgroup into an LDS patch, B fragments
   * straight from L2); NULL = conv_x3_kernel / igemm_kernel as above. */
  const void* w_frag;
  /* Optional: the dispatch options of THIS launch -- vlnce_option_count() ints indexed by
   * vlnce_option_index(name), a negative entry = the process value (vlnce_set_option); NULL = the
   * process values.  Two policies (or a test forcing a kernel) in one process then do not share
   * mutable dispatch state: the library only reads what the call hands it. */
  const int* options;
  /* Plane format of w_split / w_frag (ABI 142) = the arithmetic of the plane kernels for this
   * launch: 0 or 1 = three bf16 planes (six plane products per multiply), 2 = fp16 planes (three
   * products); must be the `format` the two buffers were made with. */
  int w_format;
} vlnce_prologue;

/* Train-mode BatchNorm statistics taken BY the convolution (torch.nn.BatchNorm2d.forward in
 * training mode on the convolution's output: resnet_encoders.py:136-139 leaves the frozen trunk's
 * BatchNorm on batch statistics).  Every workgroup adds the {sum x, sum x^2} of
 * its raw output columns to `acc` with fp64 device-scope atomics (per 32-row block: the block
 * sum in fp32, x^2 as M2_block + sum^2 / rows evaluated in fp64 -- the same function of the
 * accumulators as the tile-moment path below); vlnce_bn_finalize_sums, one workgroup, then turns
 * the sums into the pending normalisation and the running statistics and leaves `acc` zero
 * again.  Together they replace convolution + [coarsen] + finalize over thousands of tile
 * moments (9-21 us behind every convolution of the train-mode trunk) by convolution + ~4 us.
 * A kernel without this epilogue (fp32-MFMA kernel, split-K) writes its tile moments to
 * `workspace` and the library reduces them into `acc` itself: the caller sees the same thing. */
#define VLNCE_BN_SHARDS 16 /* copies of the sums: workgroup b adds to copy b % 16 (atomic contention) */
typedef struct {
  double* acc;            /* [VLNCE_BN_SHARDS][Cout][2] {sum x, sum x^2}; added to, not overwritten */
  void* workspace;        /* vlnce_conv2d_bn_workspace_bytes(d) bytes, uninitialised               */
  long workspace_bytes;
} vlnce_bn_sums;

typedef struct {
  const float* scale;     /* [Cout] per-channel multiplier or NULL (folded eval-BN gamma/sqrt(var+eps)) */
  const float* shift;     /* [Cout] per-channel add or NULL (bias / folded BN shift)                    */
  const float* residual;  /* [M,Cout] (row stride ldr) added before the activation, or NULL            */
  int ldr;
  int act;                /* VLNCE_ACT_*                                                               */
  int accumulate;         /* Y += result (after activation) instead of Y = result                      */
  /* train-mode BatchNorm: per-(M-tile, channel) partial statistics of the RAW
   * accumulator (before scale/shift): float2 {sum, M2 about the tile mean}.
   * Layout [tiles_m][Cout][2]; tiles_m from vlnce_conv2d_tiles_m(). NULL = off. */
  float* stat_partial;
  /* train-mode BatchNorm statistics added by the convolution (see vlnce_bn_sums); excludes
   * stat_partial, scale, shift, residual, act and accumulate.  NULL = off. */
  const vlnce_bn_sums* bn;
} vlnce_epilogue;

long vlnce_conv2d_bn_workspace_bytes(const vlnce_conv_desc* d); /* vlnce_bn_sums.workspace */
/* The sums of vlnce_bn_sums (all VLNCE_BN_SHARDS copies, M values per channel) -> scale_out =
 * gamma * rstd, mean_out = batch mean [, shift_out = beta - mean * scale, rstd_out]; running
 * statistics updated like torch (momentum, unbiased variance); `acc` is zero afterwards. */
int vlnce_bn_finalize_sums(double* acc, int M, int C, const float* gamma, const float* beta,
                           float eps, float momentum, float* running_mean, float* running_var,
                           float* scale_out, float* shift_out, float* mean_out, float* rstd_out,
                           vlnce_stream_t stream);

int vlnce_conv2d_tiles_m(const vlnce_conv_desc* d);   /* rows of stat_partial  */
int vlnce_conv2d_tile_rows(const vlnce_conv_desc* d); /* BM chosen for `d`      */

/* planes[q][i], q = 0..2; `planes` holds 3 * count 16-bit words.
 * format 1: bf16 words with w[i] == planes[0][i] + planes[1][i] + planes[2][i] exactly
 *   (round-to-nearest three-way split, 8 + 8 + 8 mantissa bits); a product of two operands split
 *   this way is six plane products (dropped terms <= 2^-26 relative).
 * format 2 (ABI 142): fp16 words {h * 2^11, (w - h) * 2^11 rounded, h} with h = fp16(w) -- 11 + 11
 *   mantissa bits (+ the sign of the second term: |w - h - l| <= 2^-22 |w|); the activations are
 *   split the same way in the operand loader ({a1, a2 * 2^11}) and a product is the THREE plane
 *   products a1 b1 + a1 b2 + a2 b1, all at the scale 2^11 (the low planes stay clear of fp16's
 *   subnormals; the accumulator is multiplied by 2^-11 where it leaves the registers).  Measured
 *   against fp64 the result is as close as format 1's (fp32 accumulation dominates both, and
 *   format 2 does half as many accumulator updates) at half the matrix-pipe time.  Range:
 *   |w| < 32 and |activation| < 65504, beyond that the output is inf / NaN (never a wrong finite
 *   value); operands far below 6e-5 (gradients) lose relative precision: backward launches use
 *   format 1. */
int vlnce_conv2d_split_weights(const float* w, void* planes, long count, int format,
                               vlnce_stream_t stream);

/* frag = the weights of `d` as bf16-plane MFMA B fragments, layout
 * [Cout/32][K/16][3 planes][64 lanes][8 bf16] with the k-slabs ordered (32-channel chunk, filter
 * tap, 16-channel half) and w == plane0 + plane1 + plane2 exactly (round-to-nearest split).
 * vlnce_conv2d_pack_bytes() = bytes `frag` must hold (Cout*K*6), or 0 where the fragment kernel
 * does not apply (Cin % 32 != 0 or Cout % 32 != 0).  Replaces nothing upstream: a cached
 * re-layout of nn.Conv2d.weight (resnet_encoders.py:136-139). */
long vlnce_conv2d_pack_bytes(const vlnce_conv_desc* d);
int vlnce_conv2d_pack_weights(const float* w_ohwi, void* frag, const vlnce_conv_desc* d,
                              int format /* as vlnce_conv2d_split_weights */, vlnce_stream_t stream);

/* Every weight image of a TRAINABLE trunk in one launch (ABI 143).  With
 * MODEL.{RGB,DEPTH}_ENCODER.trainable (resnet_encoders.py:45-46,141-143) the filters change every
 * optimizer step; per convolution the step needs the forward bank W[Cout,T,Cin] (T = KH*KW taps) and
 * the data-gradient bank W'[Cin,T,Cout] with the taps reversed, each as fp32, as planes
 * (vlnce_conv2d_split_weights' layout) and as fragments (vlnce_conv2d_pack_weights' layout) -- bit
 * for bit what those entry points write for the permuted tensors.  A job names ONE of these outputs
 * and reads the parameter where it lies, in nn.Conv2d's own [Cout, Cin, KH, KW] layout.
 *   kind 0: fp32 [N,T,C];  1: planes [3][N*T*C];  2: fragments (N % 32 == 0 and C % 32 == 0)
 *   with (N, C) = (Cout, Cin), or (Cin, Cout) when `transposed`; C % 8 == 0 throughout.
 * vlnce_weight_job_items() = the work items of a job (-1: not eligible).  `jobs_dev` and
 * `first_item_dev` (njobs + 1 running item counts, first_item[0] = 0) are DEVICE arrays the host
 * builds once per trunk; total_items = first_item[njobs]. */
typedef struct vlnce_weight_job {
  const float* w_oihw;
  void* dst;
  int Cout, Cin, T;
  int kind;
  int transposed;
  int format; /* plane format of kinds 1 and 2 (1 = three bf16 planes, 2 = fp16 planes) */
} vlnce_weight_job;
long vlnce_weight_job_items(const vlnce_weight_job* job);
int vlnce_conv2d_prepare_weights(const vlnce_weight_job* jobs_dev, const long* first_item_dev,
                                 int njobs, long total_items, vlnce_stream_t stream);

/* Which kernel the calling thread's last vlnce_conv2d_fwd was dispatched to: the measurement
 * harness prices bf16-pipe launches (6 plane products per multiply) and fp32-MFMA launches
 * against their own peaks.  No reference counterpart. */
#define VLNCE_CONV_PATH_F32 0 /* igemm_kernel, v_mfma_f32_32x32x2_f32 (incl. split-K)            */
#define VLNCE_CONV_PATH_X3 1  /* conv_x3_kernel, bf16 planes, im2col K-tiles through LDS          */
#define VLNCE_CONV_PATH_P3 2  /* conv_p3_kernel, bf16 planes, patch-resident A / fragment-order B */
#define VLNCE_CONV_PATH_M3 3  /* conv_m3_kernel, bf16 planes, small launches: no LDS staging, 4-way k split */
int vlnce_conv2d_last_path(void);

int vlnce_conv2d_fwd(const float* x, const float* w_ohwi, float* y,
                     const vlnce_conv_desc* d, const vlnce_prologue* pro,
                     const vlnce_epilogue* epi, vlnce_stream_t stream);

/* General GEMM  C[M,N] = act( op(A)[M,K] * op(B)[K,N] * scale[n] + shift[n] + R )
 *   transA = 0: A is [M,K] row-major (lda)     transA = 1: A is stored [K,M] (lda)
 *   transB = 0: B is [N,K] row-major (ldb)  -- i.e. an nn.Linear weight
 *   transB = 1: B is stored [K,N] row-major (ldb)
 * Replaces nn.Linear / 1x1 nn.Conv1d forward and their dgrad/wgrad
 * (cma_policy.py:103-119,140-170; seq2seq_policy.py:109-121;
 * waypoint_predictors.py:76-180; policy.py:19-21).
 */
int vlnce_gemm(const float* A, int lda, int transA, const float* B, int ldb, int transB,
               float* C, int ldc, int M, int N, int K,
               const vlnce_epilogue* epi, vlnce_stream_t stream);

/* Skinny linear layers (M <= 128 rows = one row per environment; N % 4 == 0, K % 4 == 0; rows of x,
 * w, dy, y 16-byte aligned), ABI 141: y = act(x W^T + b) in ONE launch and the whole backward --
 * dx [M,K] = dz W, dw [N,K] = dz^T x, db [N] = colsum dz with dz = dy * act'(y) -- in ONE launch
 * (each of dx / dw / db may be NULL), exact fp32 MFMA, no atomics, no pre-zeroed outputs.
 * Replace nn.Linear / the nn.GRU projections of the nets at one row per environment and their
 * autograd (cma_policy.py:103-131,140-177; seq2seq_policy.py:109-121; waypoint_predictors.py:76-180),
 * which through vlnce_gemm are 3 launches forward and 5 backward per layer. */
int vlnce_linear_rows_supported(int M, int N, int K);
int vlnce_linear_rows_fwd(const float* x, int ldx, const float* w, int ldw, const float* bias,
                          int act, float* y, int ldy, int M, int N, int K, vlnce_stream_t stream);
int vlnce_linear_rows_bwd(const float* x, int ldx, const float* w, int ldw, const float* dy,
                          int lddy, const float* y, int ldy, int act, float* dx, float* dw,
                          float* db, int M, int N, int K, vlnce_stream_t stream);

/* column sums: out[n] (+)= sum_m X[m,n]  (bias gradients) */
int vlnce_colsum(const float* x, int ldx, int M, int N, float* out, int accumulate,
                 vlnce_stream_t stream);

/* ------------------------------------------------------------- normalisation
 * BatchNorm2d (torchvision trunk; eps 1e-5, momentum 0.1; SURVEY App. B-1:
 * runs on batch statistics whenever the policy was not .eval()'d).
 * vlnce_bn_finalize reduces the conv epilogue's partials (Chan's parallel
 * variance, fp64 combine) into per-channel scale/shift and updates the running
 * statistics exactly like torch (unbiased running_var). */
size_t vlnce_bn_finalize_workspace_bytes(int tiles_m, int C); /* 0 unless tiles_m > 4096 */
int vlnce_bn_finalize(const float* stat_partial, int tiles_m, int tile_rows, int M, int C,
                      const float* gamma, const float* beta, float eps, float momentum,
                      float* running_mean, float* running_var, /* may be NULL */
                      float* scale_out, float* shift_out,
                      float* mean_out, float* rstd_out, /* may be NULL; saved for backward */
                      void* workspace, size_t workspace_bytes, /* caller-allocated scratch */
                      vlnce_stream_t stream);

/* y = act((x - center[s, c]) * scale[s, c] + shift[s, c] + residual)   with s = row / rows_per_sample
 * (rows_per_sample = 0 => one vector for all rows: BatchNorm apply; > 0 => per-sample vectors
 * [Nsamples, C]: GroupNorm apply).  center may be NULL (= 0).  Batch-statistics normalisation
 * passes center = mean, scale = gamma*rstd, shift = beta: the reference's own arithmetic
 * ((x - mean) * rstd * gamma + beta), which keeps fp32 accuracy when |mean| >> std. */
int vlnce_scale_shift_act(const float* x, const float* scale, const float* shift,
                          const float* center, int rows_per_sample, const float* residual,
                          float* y, long M, int C, int act, vlnce_stream_t stream);

/* y = act(x1*scale1[c]+shift1[c] + x2*scale2[c]+shift2[c]); y may alias x1.  End of a residual
 * block with a conv+BatchNorm downsample: both raw conv outputs normalised in one pass. */
int vlnce_scale_shift_add_act(const float* x1, const float* scale1, const float* shift1,
                              const float* center1, const float* x2, const float* scale2,
                              const float* shift2, const float* center2, float* y, long M, int C,
                              int act, vlnce_stream_t stream);

/* GroupNorm (habitat depth trunk: ngroups 16, and GroupNorm(1,C) in the
 * compression block; eps 1e-5).  Two launches: partial sums per
 * (sample, pixel-chunk, channel) then a finalize that emits per-(sample,channel)
 * scale/shift for vlnce_scale_shift_act. */
int vlnce_gn_chunks(int HW);
int vlnce_gn_partial(const float* x, int Nimg, int HW, int C, float* partial /* [N,chunks,C,2] */,
                     vlnce_stream_t stream);
int vlnce_gn_finalize(const float* partial, int Nimg, int HW, int C, int groups,
                      const float* gamma, const float* beta, float eps,
                      float* scale_out, float* shift_out, /* [N,C] */
                      float* center_out, /* [N,C] or NULL: if given, shift_out = beta and
                                            center_out = mean (unfolded form) */
                      float* mean_out, float* rstd_out,   /* [N,groups] or NULL */
                      vlnce_stream_t stream);
/* Same result from the convolution epilogue's tile statistics (vlnce_epilogue.stat_partial,
 * {sum, M2} per tile of tile_rows output pixels): usable when HW % tile_rows == 0, i.e. no tile
 * straddles two samples; saves the vlnce_gn_partial pass over the activation. */
int vlnce_gn_finalize_tiles(const float* stat_partial, int tile_rows, int Nimg, int HW, int C,
                            int groups, const float* gamma, const float* beta, float eps,
                            float* scale_out, float* shift_out, float* center_out,
                            float* mean_out, float* rstd_out, vlnce_stream_t stream);
/* y = act(GroupNorm(x) + residual) of a small activation x [N, HW, C] in ONE launch (one
 * workgroup per (sample, group): mean, centred second moment, apply): what the policy uses at
 * one to a few environments (act(), resnet_encoders.py:31-43 GroupNorm trunk), where the three
 * launches of the pipeline above are latency, not bandwidth.  residual may be NULL; y may alias x. */
int vlnce_group_norm_small(const float* x, int Nimg, int HW, int C, int groups, const float* gamma,
                           const float* beta, float eps, const float* residual, int act, float* y,
                           vlnce_stream_t stream);

/* ------------------------------------------------------------------ pooling
 * NHWC. maxpool 3x3/s2/p1 (torchvision + habitat stems), avg_pool2d(2)
 * (ResNetEncoder.forward), adaptive_avg_pool2d -> (OH,OW)
 * (resnet_encoders.py:154-162; also the global 1x1 pool). */
/* maxpool over act(x*in_scale[c]+in_shift[c]) when in_scale != NULL: the stem's BatchNorm+ReLU
 * is applied on the fly to the raw conv output (train mode), saving one HBM round trip. */
int vlnce_maxpool3x3s2(const float* x, float* y, int N, int H, int W, int C, int Ho, int Wo,
                       const float* in_scale, const float* in_shift, const float* in_center,
                       int in_relu, vlnce_stream_t stream);
int vlnce_avgpool2x2(const float* x, float* y, int N, int H, int W, int C, vlnce_stream_t stream);

/* 2x2 space-to-depth with explicit zero border and optional per-channel input transform:
 *   y[n, pb_h, pb_w, (dy*2+dx)*C + c] = x[n, 2*(pb_h-pad_lo)+dy, 2*(pb_w-pad_lo)+dx, c]*scale[c]+shift[c]
 * y is [N, H/2+pad_lo+pad_hi, W/2+pad_lo+pad_hi, 4C].  With pad_lo=2, pad_hi=1 a 7x7/stride-2/pad-3
 * stem convolution (torchvision ResNet conv1, resnet_encoders.py:136-139; habitat ResNet conv1)
 * over x equals a 4x4/stride-1/pad-0 convolution over y with the taps regrouped: the 3-channel
 * gather becomes contiguous 16-byte loads (K = 64*C instead of 49*C scattered elements).  The
 * transform (/255, ImageNet mean/std: resnet_encoders.py:184-190) is applied before the zero
 * border, exactly like the reference normalises before the padded convolution. */
int vlnce_space_to_depth2(const float* x, float* y, int N, int H, int W, int C, int pad_lo,
                          int pad_hi, const float* scale, const float* shift,
                          vlnce_stream_t stream);
/* ------------------------------------------------------------- observation ingest
 * What habitat's batch_obs (fp32 cast), CenterCropperPerSensor / ObsStack
 * (habitat_extensions/obs_transformers.py:21-145), the waypoint net's concatenation of 12
 * panorama frames with the done-masked history frame (waypoint_predictors.py:330-375) and the
 * encoders' own input handling (/255, avg_pool2d(2): resnet_encoders.py:95,198-199) do in four
 * passes over fp32 frames, in one pass over the frames in their STORAGE type (uint8 RGB). */
#define VLNCE_DT_F32 0
#define VLNCE_DT_U8 1
typedef struct {
  const void* x;              /* [N, F, Hs, Ws, C] frames (F = 1: a plain [N, Hs, Ws, C] sensor)      */
  const void* x2;             /* optional extra frame per env [N, Hs, Ws, C], appended as frame F      */
  const unsigned char* mask2; /* optional [N]: the extra frame is multiplied by it (not-done mask)     */
  int dtype;                  /* VLNCE_DT_F32 | VLNCE_DT_U8: element type of x and x2                  */
  int N, F, Hs, Ws, C;
  int y0, x0, H, W;           /* centre-crop window: rows [y0, y0+H), columns [x0, x0+W) of each frame */
} vlnce_frames;
/* RGB stem input: 2x2 space-to-depth of every (cropped) frame with the per-channel input
 * transform, y [N*(F + (x2 != NULL)), H/2+pad_lo+pad_hi, W/2+pad_lo+pad_hi, 4C] fp32; image index
 * n*(F+1) + f as torch.cat([frames, history.unsqueeze(1)], 1).flatten(0, 1) orders them. */
int vlnce_frames_s2d(const vlnce_frames* frames, float* y, int pad_lo, int pad_hi,
                     const float* scale, const float* shift, vlnce_stream_t stream);
/* RGB stem in one launch, from the frames: y[img, ho, wo, :] = conv 7x7 / stride 2 / pad 3 of
 * (frame * in_scale[c] + in_shift[c]) (zero outside the frame) with `Cout` = 32 | 64 filters --
 * torchvision ResNet.conv1 behind the encoder's /255 (+ ImageNet mean/std),
 * resnet_encoders.py:131-139,171-199 -- on the 16-bit matrix pipe (w_format 1: three exact bf16
 * planes per operand, six plane products; 2: fp16 planes, three products -- the formats of
 * vlnce_conv2d_split_weights; fp32 accumulation).  w_frag: the filters as B fragments
 * [Cout/32][11 k-slabs][3 planes][64 lanes][8 x 16 bit], k' = kh * 24 + kw * 3 + c (21 of every 24
 * used, the rest and k' >= 168 zero), lane (l, h) = output channel 32 nb + l, k' = 16 ks + 8 h + [0, 8).
 * epi: NULL, {scale, shift, act} (eval: folded BatchNorm + ReLU) or {bn} (train: raw output, the
 * launch adds its column sums to bn->acc; workspace unused). */
int vlnce_stem7_fwd(const vlnce_frames* frames, const float* in_scale, const float* in_shift,
                    const void* w_frag, int w_format, float* y, int Cout, const vlnce_epilogue* epi,
                    vlnce_stream_t stream);
/* depth stem input: F.avg_pool2d(x, 2) of every frame, y [N*(F+..), H/2, W/2, C] fp32 */
int vlnce_frames_avgpool2(const vlnce_frames* frames, float* y, vlnce_stream_t stream);
/* the frames as fp32 [N*(F+..), H, W, C] (x*scale+shift when given): stems that cannot take the
 * space-to-depth form, trainable encoders */
int vlnce_frames_f32(const vlnce_frames* frames, float* y, const float* scale, const float* shift,
                     vlnce_stream_t stream);
/* eager ObsStack + CenterCropperPerSensor: out[n, f, h, w, :] = srcs[f][n, y0+h, x0+w, :] for F
 * (<= 16) source sensors [N, Hs, Ws, C] of elem_bytes-wide elements; `srcs` is a HOST array of
 * device pointers; out [N, F, H, W, C] in the source element type (bytes are moved as they are). */
int vlnce_frames_gather(const void* const* srcs, int F, int elem_bytes, int N, int Hs, int Ws,
                        int C, int y0, int x0, int H, int W, void* out, vlnce_stream_t stream);

/* habitat's ResizeShortestEdge [3P, habitat-lab v0.1.7 habitat_baselines/common/obs_transformers.py;
 * enabled by every RxR config, rxr_baselines/rxr_cma_en.yaml:27-30, applied at
 * base_il_trainer.py:284-285] = image_resize_shortest_edge = F.interpolate(mode="area") to
 * (OH, OW) = (int(Hs*size/min(Hs,Ws)), int(Ws*size/min(Hs,Ws))), cast back to the sensor dtype --
 * fused with the CenterCropperPerSensor window that follows it:
 *   out[i, h, w, c] = cast(mean of x[i, hs0:hs1, ws0:ws1, c]),  hs0 = floor((y0+h)*Hs/OH),
 *   hs1 = ceil((y0+h+1)*Hs/OH), columns likewise (at::adaptive_avg_pool2d's windows and
 *   summation order; uint8 truncates).  x [NF, Hs, Ws, C] and out [NF, H, W, C] in `dtype`
 *   (VLNCE_DT_U8 | VLNCE_DT_F32); (y0, x0, H, W) = (0, 0, OH, OW) is the plain resize. */
int vlnce_frames_resize_area(const void* x, int dtype, int NF, int Hs, int Ws, int C, int OH, int OW,
                             int y0, int x0, int H, int W, void* out, vlnce_stream_t stream);

int vlnce_adaptive_avgpool(const float* x, float* y, int N, int H, int W, int C, int OH, int OW,
                           int ldy, vlnce_stream_t stream);

/* ------------------------------------------------- backward of the visual trunks
 * (only reached with MODEL.RGB_ENCODER / DEPTH_ENCODER .trainable = True; the reference
 * default keeps both encoders frozen).  Data gradients of stride-1 convolutions reuse
 * vlnce_conv2d_fwd with the flipped / transposed weights. */
/* dW[Cout,KH,KW,Cin] = sum_m dY[m,co] * im2col(X)[m,(r,q,ci)]   (split over output pixels, fp32 atomics
 * into dW, which the call zeroes itself).  Cin % 32 == 0 and Cout % 32 == 0: six bf16-plane products
 * per multiply on the 16-bit matrix pipe (plane format 1: the operands are gradients); else, and with
 * option "wgrad_tile" = 1, v_mfma_f32_32x32x2_f32.
 * dy_pow2 / P (ABI 143, may be NULL / 0): the [2][P] power-of-two buffer vlnce_bn_bwd / vlnce_gn_bwd filled
 * for this dy (dy_pow2[0] = 2^k, dy_pow2[P] = 2^-k) -- with it the plane kernel runs in format 2:
 * x on the two-plane side (|x| < 65504), dy * 2^(k-10) on the three-plane side, three plane
 * products per multiply, the result scaled back exactly.
 * accumulate != 0 (ABI 143): dW += instead of dW = (the call does not zero dW). */
int vlnce_conv2d_wgrad(const float* x, const float* dy, float* dw_ohwi, const vlnce_conv_desc* d,
                       const float* dy_pow2, int P, int accumulate, vlnce_stream_t stream);
/* BatchNorm2d backward through y = act(x*gamma*rstd + (beta - mean*gamma*rstd) (+ residual)):
 * g = dy*[y>0] when relu; dbeta = sum g; dgamma = sum g*xhat;
 * dx = gamma*rstd*(g - dbeta/M - xhat*dgamma/M) with batch statistics, gamma*rstd*g with
 * running statistics (use_batch_stats = 0).  dres (may be NULL) receives g.  workspace (ABI 143):
 * vlnce_bn_bwd_workspace_floats(M, C) floats -- the per-block partial sums of the two reductions
 * (no same-address atomics); 0 floats / may be NULL when C % 4 != 0.
 * pow2 (ABI 143, may be NULL; needs C % 4 == 0): [2][P] floats, on return P copies of 2^k followed by P
 * copies of 2^-k, with k the power of two at which dx fits the fp16 planes of plane format 2
 * (bound(|dx|) * 2^k in (2^13, 2^14], the bound computed from the per-channel maxima of |g| and
 * |x - mean| gathered by the reduction pass).  They are the prologue / epilogue vectors
 * (vlnce_prologue.in_scale with a zero in_shift; vlnce_epilogue.scale) of the data-gradient
 * convolution that reads dx, and the `dy_pow2` of vlnce_conv2d_wgrad: gradients live far below
 * fp16's normal range, scaled by an exact power of two they take format 2's three plane products
 * instead of format 1's six. */
size_t vlnce_bn_bwd_workspace_floats(long M, int C);
int vlnce_bn_bwd(const float* dy, const float* y, const float* x, const float* mean,
                 const float* rstd, const float* gamma, long M, int C, int relu,
                 int use_batch_stats, float* dx, float* dres, float* dgamma, float* dbeta,
                 float* workspace, float* pow2, int P, vlnce_stream_t stream);
/* GroupNorm backward (same conventions; mean/rstd are [N,groups]); workspace from
 * vlnce_gn_bwd_workspace_floats() floats; pow2 / P as vlnce_bn_bwd. */
size_t vlnce_gn_bwd_workspace_floats(int Nimg, int HW, int C, int groups);
int vlnce_gn_bwd(const float* dy, const float* y, const float* x, const float* mean,
                 const float* rstd, const float* gamma, int Nimg, int HW, int C, int groups,
                 int relu, float* dx, float* dres, float* dgamma, float* dbeta, float* workspace,
                 float* pow2, int P, vlnce_stream_t stream);
/* max-pool forward that also records the arg-max tap (0..8), and its backward */
int vlnce_maxpool3x3s2_argmax(const float* x, float* y, uint8_t* argmax, int N, int H, int W,
                              int C, int Ho, int Wo, vlnce_stream_t stream);
int vlnce_maxpool3x3s2_bwd(const float* dy, const uint8_t* argmax, float* dx, int N, int H, int W,
                           int C, int Ho, int Wo, vlnce_stream_t stream);
int vlnce_adaptive_avgpool_bwd(const float* dy, float* dx, int N, int H, int W, int C, int OH,
                               int OW, vlnce_stream_t stream);

/* ---------------------------------------------------------------- categorical action head
 * habitat-lab CategoricalNet (built at models/policy.py:19-21; the distribution is utils.py:269-289):
 *   z = x W^T + b;  torch.distributions.Categorical(logits=z) keeps  z - logsumexp(z, -1)  and its
 *   argument validation raises when that holds a NaN.
 * fwd: x [M,K] (row stride ldx), w [A,K], b [A] or NULL, A <= 16  ->  logits_out [M,A] (normalised);
 *      *nan_count (device int, or NULL) is incremented once per row that holds a NaN -- the host
 *      reads it where the reference's validation reads its own flag.
 * bwd: logits = what fwd returned, dlogits = gradient w.r.t. it;  dx [M,K], dw [A,K], db [A]
 *      (each may be NULL; db needs dw).  One launch (plus a zero-fill when M > 1024, where the rows
 *      of dw are reduced by several workgroups). */
int vlnce_action_head_fwd(const float* x, int ldx, const float* w, const float* b, int M, int K,
                          int A, float* logits_out, int* nan_count, vlnce_stream_t stream);
int vlnce_action_head_bwd(const float* x, int ldx, const float* w, const float* logits,
                          const float* dlogits, int M, int K, int A, float* dx, float* dw,
                          float* db, vlnce_stream_t stream);

/* ---------------------------------------------------------------- attention
 * One query per batch row against P keys:  logits[i] = <q, K[i]>;
 *   mask_mode 1 (additive, CMANet._attn cma_policy.py:207-217): logits -= mask*1e8
 *   mask_mode 2 (multiplicative, DotProductAttention utils.py:173-177): logits *= mask
 *   attn = softmax(logits * scale);  out = sum_i attn[i] * V[i]
 * K: [B,P,Dk] (row stride ldk), V: [B,P,Dv] (row stride ldv), mask: uint8 [B,P] or NULL.
 * attn_out [B,P] is saved for the backward.  P <= 1024. */
int vlnce_attn_fwd(const float* q, const float* K, int ldk, const float* V, int ldv,
                   const uint8_t* mask, int mask_mode, float scale,
                   float* out, float* attn_out, int B, int P, int Dk, int Dv,
                   vlnce_stream_t stream);
int vlnce_attn_bwd(const float* dout, const float* q, const float* K, int ldk,
                   const float* V, int ldv, const uint8_t* mask, int mask_mode, float scale,
                   const float* attn, float* dq, float* dK, int lddk, float* dV, int lddv,
                   int B, int P, int Dk, int Dv, vlnce_stream_t stream);

/* The same attention with K / V / mask SHARED by groups of queries: query row b attends over block
 * kv_index[b] (0 <= kv_index[b] < U) of K [U,P,Dk], V [U,P,Dv], mask [U,P].  A sequence-mode batch
 * (T*N rows: a cached-feature DAgger batch, dagger_trainer.py:39-114) repeats every episode's
 * instruction T times; the instruction encoder runs once per DISTINCT instruction and the text
 * attention (cma_policy.py:260) reads the distinct blocks in place instead of a T-fold expanded
 * copy.  bwd: dq [B,Dk]; dK [B,P,Dk] / dV [B,P,Dv] are PER QUERY ROW (as vlnce_attn_bwd writes
 * them) -- vlnce_segment_sum reduces them to the U blocks. */
int vlnce_attn_fwd_shared(const float* q, const float* K, int ldk, const float* V, int ldv,
                          const uint8_t* mask, int mask_mode, float scale, const int64_t* kv_index,
                          float* out, float* attn_out, int B, int P, int Dk, int Dv,
                          vlnce_stream_t stream);
int vlnce_attn_bwd_shared(const float* dout, const float* q, const float* K, int ldk,
                          const float* V, int ldv, const uint8_t* mask, int mask_mode, float scale,
                          const int64_t* kv_index, const float* attn, float* dq, float* dK,
                          int lddk, float* dV, int lddv, int B, int P, int Dk, int Dv,
                          vlnce_stream_t stream);
/* out[u, :] = sum of the rows b of x [B, row_elems] with index[b] == u, added in row order
 * (deterministic; the backward of a row gather `x_u.index_select(0, index)`).  row_elems % 4 == 0,
 * U <= 65535. */
int vlnce_segment_sum(const float* x, const int64_t* index, int B, int U, long row_elems,
                      float* out, vlnce_stream_t stream);
/* mask[b,i] = all_c(x[b,i,c] == 0)   (text_mask, cma_policy.py:260) */
int vlnce_rowzero_mask(const float* x, int ld, long rows, int C, uint8_t* mask,
                       vlnce_stream_t stream);

/* --------------------------------------------------------- recurrent cells
 * GRU / LSTM pointwise gate stage (torch.nn.GRU/LSTM semantics, gate order
 * r,z,n / i,f,g,o) with the done-mask already applied to h_prev by the caller
 * through `mask` (h_prev_eff = h_prev * mask[b]); gi = x W_ih^T + b_ih and
 * gh = h_prev_eff W_hh^T + b_hh come from vlnce_gemm.  Replaces habitat's
 * RNNStateEncoder single/seq forward (seq2seq_policy.py:164, cma_policy.py:
 * 249-256,287-294, waypoint_predictors.py:420-427,532-545). */
int vlnce_gru_gates_fwd(const float* gi, const float* gh, const float* h_prev,
                        const uint8_t* mask, /* [B] or NULL */
                        float* h_out, float* gates_out /* [B,3H] r,z,n saved */,
                        float* hn_out /* [B,H] saved W_hn h + b_hn */,
                        int B, int H, vlnce_stream_t stream);
int vlnce_gru_gates_bwd(const float* dh_out, const float* gates, const float* hn,
                        const float* h_prev, const uint8_t* mask,
                        float* dgi, float* dgh, float* dh_prev, int B, int H,
                        vlnce_stream_t stream);
int vlnce_lstm_gates_fwd(const float* gi, const float* gh, const float* c_prev,
                         const uint8_t* mask, float* h_out, float* c_out,
                         float* gates_out /* [B,4H] i,f,g,o activated */, int B, int H,
                         vlnce_stream_t stream);
int vlnce_lstm_gates_bwd(const float* dh_out, const float* dc_out, const float* gates,
                         const float* c_prev, const float* c_out, const uint8_t* mask,
                         float* dgates /* [B,4H] */, float* dc_prev, int B, int H,
                         vlnce_stream_t stream);

/* Packed-sequence instruction RNN (instruction_encoder.py:27-32,80-94): the whole
 * time loop (and its BPTT) in ONE launch, one workgroup per (direction, 16-sample
 * tile), recurrent weights resident in registers, h W_hh^T on v_mfma_f32_16x16x4_f32.
 * kind 0 = LSTM (gates i|f|g|o), 1 = GRU (r|z|n); dirs 1 or 2 (index 1 = reverse).
 * All per-direction arguments are HOST arrays of `dirs` device pointers.
 *   gi[d]      [L,B,G*H]  x W_ih^T + b_ih, time-major
 *   w_hh[d]    [G*H,H], b_hh[d] [G*H]
 *   out[d]     [L,B,H]    must be pre-zeroed: steps t >= lengths[b] emit zeros
 *   h_final[d] [B,H]      state after each sample's last step
 *   gates_save[d] [L,B,G*H], aux_save[d] [L,B,H] (LSTM: c_t; GRU: W_hn h + b_hn) or NULL arrays
 * Steps t >= lengths[b] leave the state untouched; the reverse direction walks
 * t = lengths[b]-1 .. 0 (pack_padded_sequence / pad_packed_sequence semantics). */
int vlnce_rnn_seq_supported(int kind, int H);
int vlnce_rnn_seq_fwd(int kind, int dirs, const float* const* gi, const float* const* w_hh,
                      const float* const* b_hh, const int* lengths, float* const* out,
                      float* const* h_final, float* const* gates_save, float* const* aux_save,
                      int B, int L, int H, vlnce_stream_t stream);
/* BPTT: w_hh_t[d] = W_hh^T [H,G*H]; dout[d] [L,B,H] / dh_final[d] [B,H] may be NULL;
 * dgi[d] [L,B,G*H] (pre-zeroed) receives d/d gi; GRU additionally fills dgh[d]
 * (gradient wrt h W_hh^T + b_hh; its n-gate differs by the factor r). */
int vlnce_rnn_seq_bwd(int kind, int dirs, const float* const* w_hh_t, const int* lengths,
                      const float* const* out, const float* const* gates_save,
                      const float* const* aux_save, const float* const* dout,
                      const float* const* dh_final, float* const* dgi, float* const* dgh,
                      int B, int L, int H, vlnce_stream_t stream);
/* Second-generation entry points (ABI 141): the same two kernels, self-contained -- the launch
 * writes the zeros past each row's length itself (no pre-zeroed buffers), the forward also stores
 * the outputs in the CONSUMER's layout (`seq`, may be NULL: element (t, b, direction d, unit u) at
 * seq[t * seq_st + b * seq_sb + d * H + u]; [B, L, dirs*H] rows = the reference's
 * pad_packed_sequence output permuted, instruction_encoder.py:88-94, with seq_st = dirs*H,
 * seq_sb = L*dirs*H) next to the time-major copy `out_tm[d]` [L,B,H] the backward needs, BPTT takes
 * the output gradient in that same addressing (`dseq`, may be NULL; regrouped into `dout_ws` by a
 * small launch of the same call) and W_hh as the module stores it ([G*H, H], no transposed copy).
 * dgi / dgh [L,B,G*H] are written everywhere. */
int vlnce_rnn_seq_fwd2(int kind, int dirs, const float* const* gi, const float* const* w_hh,
                       const float* const* b_hh, const int* lengths, float* const* out_tm,
                       float* seq, long seq_st, long seq_sb, float* const* h_final,
                       float* const* gates_save, float* const* aux_save, int B, int L, int H,
                       vlnce_stream_t stream);
int vlnce_rnn_seq_bwd2(int kind, int dirs, const float* const* w_hh, const int* lengths,
                       const float* const* out_tm, const float* const* gates_save,
                       const float* const* aux_save, const float* dseq, long dseq_st, long dseq_sb,
                       float* dout_ws /* dirs*L*B*H floats, needed with dseq */,
                       const float* const* dh_final, float* const* dgi, float* const* dgh,
                       int B, int L, int H, vlnce_stream_t stream);
/* Every parameter gradient of the recurrent layer, and the gradient of its input rows, from what
 * BPTT left in dgi / dgh, behind ONE call (torch: the autograd of nn.LSTM / nn.GRU's weight_ih,
 * weight_hh, bias_ih, bias_hh): dw_hh[d] [G*H,H] = dGh^T Hprev, db_hh[d] = colsum dGh (dGh = dgh
 * for a GRU, dgi for an LSTM), dw_ih[d] [G*H,E] = dgi^T X, db_ih[d] = colsum dgi (skipped when
 * db_ih[d] == db_hh[d]: an LSTM's two bias gradients are equal), dx_tm [L*B, E] (may be NULL) =
 * sum_d dgi[d] W_ih[d].  x_tm: the time-major input rows [L*B, E] (row stride ldx).  first_dir: the
 * direction element 0 of the arrays is (0 forward, 1 reverse: a call for one direction alone, so that
 * a host can run the two directions on two streams). */
int vlnce_rnn_seq_wgrad(int kind, int dirs, int first_dir, const float* const* dgi, const float* const* dgh,
                        const float* const* out_tm, const float* x_tm, int ldx, int E,
                        const float* const* w_ih, float* const* dw_ih, float* const* dw_hh,
                        float* const* db_ih, float* const* db_hh, float* dx_tm, int B, int L, int H,
                        vlnce_stream_t stream);

/* ---------------------------------------------------------------- utilities */
/* y[b, c] = mean_p x[b, p, c]   (AdaptiveAvgPool1d(1) of rgb_linear, cma_policy.py:104) */
int vlnce_mean_rows(const float* x, float* y, int B, int P, int C, vlnce_stream_t stream);
/* x[b, :] *= mask[b]  (done-mask zeroing of the recurrent state; out may alias x) */
/* One step of a T-step state-encoder rollout over N <= 16 episodes, fused (habitat
 * RNNStateEncoder.seq_forward, call sites cma_policy.py:249-256,287-294; rollout_storage.py:154-276):
 *   fwd: hp = mask * h_prev; gates(gi + hp W_hh^T + b_hh); h (, c)          -- one launch
 *   bwd: dh = dout + carry; gate gradients (dgi, dgh);
 *        carry <- mask * (dh*z + dgh W_hh)  (GRU) | mask * (dgates W_hh)  (LSTM) -- two launches
 * w_hh_t is W_hh^T [H, G*H]; carry is read and overwritten; acc0 is [N,H] scratch.
 * aux = W_hn h + b_hn (GRU) | c_t (LSTM); gates = post-activation gates, torch order. */
int vlnce_rnn_step_supported(int N, int H, int lstm);
int vlnce_rnn_step_fwd(int lstm, const float* gi, const float* h_prev, const float* c_prev,
                       const uint8_t* mask, const float* w_hh, const float* b_hh, float* hp_out,
                       float* h_out, float* aux_out, float* gates_out, int N, int H,
                       vlnce_stream_t stream);
int vlnce_rnn_step_bwd(int lstm, const float* dout, float* carry, const float* dc,
                       const float* gates, const float* aux, const float* hp, const float* c_prev,
                       const uint8_t* mask, const float* w_hh_t, float* dgi, float* dgh,
                       float* acc0, float* dc_prev, int N, int H, vlnce_stream_t stream);
int vlnce_mask_rows(const float* x, const uint8_t* mask, float* out, int B, int H,
                    vlnce_stream_t stream);
/* The WHOLE T-step rollout of the masked GRU state encoder as one launch forward and one backward
 * (habitat RNNStateEncoder.seq_forward: h is multiplied by the not-done mask of step t before
 * step t; call sites cma_policy.py:249-256,287-294, seq2seq_policy.py:128-136) for N <= 16
 * episodes and H in {64,128,256,512}: H/16 workgroups keep their rows of W_hh in registers for
 * all T steps and hand each other the new state (backward: the gate gradients) as (value, step
 * tag) pairs, one 64-bit device-scope atomic store / polling load each: no fence, no counter.  Same arithmetic (fp32 FMA) and the same saved tensors as T calls of
 * vlnce_rnn_step_fwd / _bwd:
 *   fwd: gi [T,N,3H] (x W_ih^T + b_ih), h0 [N,H], mask [T,N] -> hp [T,N,H] (mask_t * h_{t-1}),
 *        out [T,N,H], gates [T,N,3H] (r, z, n), aux [T,N,H] (W_hn hp + b_hn);
 *   bwd: dout [T,N,H] (NULL = zeros), dh_final [N,H] (NULL = zeros), w_hh_t = W_hh^T [H,3H]
 *        -> dgi, dgh [T,N,3H], dh0 [N,H] (gradient wrt h0, the step-0 mask applied).
 * workspace: vlnce_gru_rollout_workspace_bytes(N, H) bytes of device memory owned by the caller
 * for the duration of the launch (the double-buffered exchange area of (value, step tag) pairs;
 * zeroed by the call).  All H/16 workgroups must become resident: do not run it beside a kernel
 * that holds every CU for longer than a few seconds (a lost workgroup traps). */
int vlnce_gru_rollout_supported(int N, int H);
long vlnce_gru_rollout_workspace_bytes(int N, int H);
int vlnce_gru_rollout_fwd(const float* gi, const float* h0, const uint8_t* mask, const float* w_hh,
                          const float* b_hh, float* hp, float* out, float* gates, float* aux,
                          void* workspace, int T, int N, int H, vlnce_stream_t stream);
int vlnce_gru_rollout_bwd(const float* dout, const float* dh_final, const float* gates,
                          const float* aux, const float* hp, const uint8_t* mask,
                          const float* w_hh_t, float* dgi, float* dgh, float* dh0,
                          void* workspace, int T, int N, int H, vlnce_stream_t stream);

/* out[b, :] = mask[b] ? a[b, :] : b[b, :]  (NULL operand = zeros): packed-sequence
 * semantics of the instruction RNN (steps past a sample's length keep the state
 * and emit zeros, instruction_encoder.py:80-94). */
int vlnce_select_rows(const uint8_t* mask, const float* a, const float* b, float* out, int B, int H,
                      vlnce_stream_t stream);
/* Backward of nn.Embedding (instruction_encoder.py:41-45,72-73: the learned token embedding):
 * grad_weight[tokens[r], :] += grad_rows[r, :] for rows with tokens[r] != padding_idx, fp32
 * atomics into a grad_weight the caller has zeroed (one launch instead of torch's sort /
 * segment / scatter chain).  tokens int64 [rows], grad_rows [rows, E], grad_weight [vocab, E]. */
int vlnce_embedding_bwd(const long* tokens, const float* grad_rows, float* grad_weight, long rows,
                        int E, long padding_idx, long vocab, vlnce_stream_t stream);
/* dz = dy * act'(.) written through the activation output y (ReLU/Sigmoid/Tanh backward). */
int vlnce_act_bwd(const float* dy, const float* y, float* dz, long n, int act,
                  vlnce_stream_t stream);

/* ---- cached-feature DAgger data path ----------------------------------
 * Replaces dagger_trainer.py:39-114 `collate_fn` (+ the .to(device, float32) that follows it at
 * :562-571) and the inflection weights of IWTrajectoryDataset.__next__ (:196-208): B ragged
 * trajectories, rows of all of them concatenated ([sum T_b, D], fp32 or fp16 as the LMDB cache
 * stores them) and an offsets vector [B+1] on the device, become the padded time-major batch
 * [Tmax*B, D] (row t*B + b) in one pass.  Observations are padded with 1.0 (sic, :77), actions
 * and weights with 0. */
enum { VLNCE_SRC_F32 = 0, VLNCE_SRC_F16 = 1, VLNCE_SRC_I64 = 2 };
/* observation sensors: any source type -> fp32 (upstream casts every sensor, tokens included) */
int vlnce_ragged_pad_rows(const void* src, int src_dtype, const int* offsets, int B, int Tmax,
                          long D, float fill, float* dst, vlnce_stream_t stream);
int vlnce_ragged_pad_rows_i64(const int64_t* src, const int* offsets, int B, int Tmax, long D,
                              int64_t fill, int64_t* dst, vlnce_stream_t stream);
/* corrected_actions [Tmax,B] (0 past the end), inflection weights [Tmax,B] (coef at t = 0 and
 * where the oracle action changes, 1 elsewhere, 0 past the end), not_done_masks [Tmax,B]
 * (0 in row t = 0, 1 elsewhere -- also past the end, as upstream). */
int vlnce_dagger_targets(const int64_t* oracle_actions, const int* offsets, int B, int Tmax,
                         float inflection_coef, int64_t* corrected_out, float* weights_out,
                         uint8_t* masks_out, vlnce_stream_t stream);

/* DD-PPO returns: RolloutStorage.compute_returns (rollout_storage.py:127-152).
 * rewards [T,N]; value_preds, masks, returns [T+1,N]; next_value [N].  GAE: value_preds[T] is set
 * to next_value and returns[0..T) = gae + value_preds; otherwise returns[T] = next_value and the
 * discounted sum runs backwards. */
int vlnce_ppo_returns(const float* rewards, float* value_preds, const float* masks,
                      const float* next_value, float* returns, int T, int N, float gamma, float tau,
                      int use_gae, vlnce_stream_t stream);

/* WDDPPO minibatch loss (ddppo_alg.py:78-121: entropy terms, clipped surrogate,
 * clipped value loss, offset L1 term) and its gradients in ONE launch (ABI 141).  All inputs [B]
 * (one value per rollout row); radians may be NULL (no "offset" action component).  stats[8] =
 * {loss, value_loss, action_loss, entropy_loss, mean pano / offset / distance entropy, offset_loss};
 * grads [5][B] = d loss / d {values, logp, ent_pano, ent_offset, ent_distance} for a unit upstream
 * gradient (torch's tie rules for max / min / clamp). */
int vlnce_ppo_loss(const float* values, const float* returns, const float* value_preds,
                   const float* logp, const float* old_logp, const float* adv,
                   const float* ent_pano, const float* ent_offset, const float* ent_distance,
                   const float* radians, int B, float clip, float value_coef, float entropy_coef,
                   float pano_coef, float offset_coef, float distance_coef, float reg_coef,
                   int use_clipped, float* stats, float* grads, vlnce_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VLNCE_HIP_H */

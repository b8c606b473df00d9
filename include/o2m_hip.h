/*
 * o2m_hip.h -- C ABI of libo2m_hip.so: the MI355X (gfx950) kernels behind the one-to-many
 * GAN training step.
 *
 * The reference (struan-robertson/one-to-many-gan) has no FFI layer of its own: its
 * "operator API" is torch.nn.functional.  Every entry point below therefore names the
 * torch call site in the reference that it replaces (paths relative to the reference
 * tree).  The Python host (one_to_many_gan_amd/_hip.py) binds these with ctypes; see
 * INTEGRATION.md for the stub a maintainer of the reference would add.
 *
 * Conventions
 *  - Plain pointers and sizes only; no torch types.  All pointers are DEVICE pointers
 *    (HBM), owned by the caller for the duration of the enqueue.  Nothing is allocated,
 *    nothing is synchronised, no global mutable state: every call only enqueues work on
 *    `stream` (a hipStream_t passed as void*), so calls are re-entrant (autograd worker
 *    threads) and capturable into a hipGraph.
 *  - Return value: 0 on success, otherwise a hipError_t (launch failure) or
 *    O2M_ERR_* (argument rejected before any launch).
 *  - Activation tensors are NHWC ("channels last"), C a multiple of 8, element type
 *    `dtype`: O2M_BF16 (bf16 storage, one bf16 MFMA per product, fp32 accumulate) or
 *    O2M_F32 (fp32 storage; the MFMA runs the bf16x3 split hi*hi + hi*lo + lo*hi with
 *    fp32 accumulate, ~2^-16 relative error per product: the parity mode).
 *  - Per-sample / per-channel vectors (scales, biases, statistics) are always fp32.
 */
#ifndef O2M_HIP_H
#define O2M_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define O2M_BF16 0
#define O2M_F32 1
/* BASELINE config #5 (fp8 weight / activation path), o2m_conv2d_fwd only: x holds OCP fp8 bytes -- e4m3 for
 * activations, e5m2 for gradients (the data-gradient call) --, w is always e4m3, y / residual are bf16. */
#define O2M_FP8_E4M3 2
#define O2M_BF8_E5M2 3

#define O2M_ACT_NONE 0
#define O2M_ACT_RELU 1
#define O2M_ACT_LRELU 2 /* negative slope 0.2 */
#define O2M_ACT_TANH 3

#define O2M_PAD_ZERO 0
#define O2M_PAD_REFLECT 1

#define O2M_ERR_BAD_ARG 10001
#define O2M_ERR_UNSUPPORTED 10002

/* ABI version; bumped whenever a struct or signature changes. */
int o2m_abi_version(void);

/* ------------------------------------------------------------------------------------
 * Per-kernel launch timing for bench.py's roofline object (measurement only; the reference
 * has no counterpart).  While enabled, every MFMA conv launch (igemm, its tail launch, wgrad,
 * the wgrad slab sum) is bracketed by a HIP-event pair on ITS launch stream and filed under
 * the kernel's name as rocprofv3 would resolve it, e.g. "conv_igemm_p8<bf16,256x256>",
 * "conv_igemm<bf16,256x64,in_scale=0>", "conv_wgrad<bf16,co128xk128>", "wgrad_reduce".
 *   o2m_launch_timing(1|0)   switch recording on / off; returns the previous state.
 *   o2m_launch_timing_read   waits for the recorded events, sums them per kernel name into
 *                            out[0..capacity) (launches, milliseconds, algorithmic flops of
 *                            the rows each launch covered), clears the records and returns
 *                            the number of kernels written (< 0: error).
 * Costs two event records per launch while on; nothing but a flag test while off. */
typedef struct o2m_launch_stat {
  char kernel[64];
  int32_t launches;
  float ms;
  double flops;
} o2m_launch_stat;
int32_t o2m_launch_timing(int32_t enable);
int32_t o2m_launch_timing_read(o2m_launch_stat* out, int32_t capacity);

/* Test hook (tests/test_hip_parity.py; never called by the product path): the number of workgroups the tile
 * selection of o2m_conv2d_fwd treats as "one per CU" (default 256: the phase-pipelined 256 x 256 kernel needs that
 * many tiles, the halo kernel twice as many, ...).  With a small value the parity cases the CPU oracle finishes in
 * seconds (gen64_deep, steps128: a handful of tiles) are ROUTED to the kernels the full-size step selects, so those
 * kernels -- not the generic tiles -- are what the oracle and the reference fixtures check.  n <= 0 restores the
 * default; returns the previous value.  No reference counterpart (the reference has no tile selection). */
int32_t o2m_debug_fill_blocks(int32_t n);

/* ------------------------------------------------------------------------------------
 * Implicit-GEMM convolution on MFMA (v_mfma_f32_32x32x16_bf16), stride 1, dilation 1.
 *   y[b,oy,ox,o] = act( out_scale[b,o] * sum_{kh,kw,i} w[o,kh,kw,i] *
 *                        (in_scale[b,i] * xpad[b,oy+kh-pad,ox+kw-pad,i]) + bias[o] )
 *                  + residual[b,oy,ox,o]
 * Replaces F.conv2d at layers.py:84-100 (EqualisedConv2d), the grouped per-sample
 * F.conv2d of Conv2dWeightModulate at layers.py:145-182 (as activation modulation:
 * in_scale = style s[b,i], out_scale = demodulation rsqrt(...)[b,o], shared weights),
 * the ReflectionPad2d feeding it (blocks.py:21,25,49,54; builder.py:162,202) via
 * pad_mode, and the activation behind it (builder.py:196,204,270,301).  The same entry
 * computes the data gradient ("convT2d") when given the flipped/transposed filter and
 * pad' = K-1-pad.
 * Ho = (H + 2*pad - KH)/stride + 1 (likewise Wo).  Ci % 8 == 0, Co % 8 == 0.
 * w layout: [Co][KH][KW][Ci] (reduction index contiguous), element type `dtype`.
 */
typedef struct {
  const void* x;          /* [B][H][W][Ci]                       */
  const void* w;          /* [Co][KH][KW][Ci]                    */
  void* y;                /* [B][Ho][Wo][Co]                     */
  const float* in_scale;  /* [B][Ci] or NULL                     */
  const float* out_scale; /* [B][Co] or NULL                     */
  const float* bias;      /* [Co]    or NULL                     */
  const void* residual;   /* [B][Ho][Wo][Co] or NULL             */
  int32_t B, H, W, Ci, Co, KH, KW, pad, pad_mode, act, dtype;
  int32_t w_batch_stride; /* 0: one filter for all samples.  >0: sample b uses w + b*stride
                             elements (pre-modulated per-sample filters, o2m_modulate_weights);
                             requires Ho*Wo % 256 == 0 so no MFMA tile straddles samples */
  int32_t stride;         /* 0 or 1: stride 1; s > 1: Ho = (H + 2*pad - KH)/s + 1.  Used by the host
                             for the space-to-depth form of the N = 3 image conv (a 7x7 stride-1
                             conv with 3 outputs = a 10x10 stride-4 conv with 48 outputs) */
  int32_t stats_mode;     /* what `stats` receives: O2M_STATS_MOMENTS (0) or O2M_STATS_DOT (1), below */
  int32_t fold_pad;       /* f > 0 (bf16 only): the conv is the data gradient of a conv behind ReflectionPad2d(f)
                             (blocks.py:17-26, builder.py:160,200): its output domain Ho x Wo is the PADDED map, and y
                             / residual are the CROPPED map [B][Ho - 2f][Wo - 2f][Co].  Output pixel (oy, ox) is ADDED
                             to y at (R(oy - f), R(ox - f)), R = the mirror the pad used (-i -> i, H-1+i -> H-1-i):
                             the adjoint of the pad, without the padded gradient ever being stored or folded by a
                             second pass.  Pixels that receive more than one contribution (rows / columns 1..f and
                             their mirror images) are zeroed by the launch and written with packed bf16 atomic adds
                             (run-to-run differences in the last bit there: not for the deterministic mode -- use
                             o2m_fold_scale_dot on the padded gradient instead); all others are plain stores.
                             `residual` is added once, by the interior contribution.  Requires act NONE, no stats,
                             stride 1, Ho - 2f >= 2f + 2 and Wo - 2f >= 2f + 2. */
  int32_t reserved1;      /* 0 */
  float* stats;           /* NULL, or InstanceNorm partial sums emitted by the epilogue:
                             stats[(m / R) * Co * 2 + o * 2 + {0, 1}] = sum / sum of squares of y[., o] (fp32,
                             before the rounding to `dtype`) over the R consecutive output pixels m .. m+R-1,
                             R = o2m_conv2d_stats_rows(d).  Requires R > 0, act NONE, no residual.  Every
                             (m / R, o) entry is written by exactly one workgroup (deterministic, no atomics);
                             o2m_instnorm_finalize turns the Ho*Wo/R partials of a sample into mean / rstd, so
                             the separate statistics pass over y (nn.InstanceNorm2d, builder.py:164,172,...;
                             blocks.py:23,27) disappears. */
  const float* deq_scale; /* fp8 dtypes only (else NULL): DEVICE pointer to four floats, the `deq` pairs
                             {1 / scale, amax} that o2m_quantize_fp8 wrote for x ([0], [1]) and for w ([2], [3]);
                             the accumulator is multiplied by [0] * [2] before out_scale / bias / activation.
                             Scales stay on the device: no host sync between quantisation and the convolution. */
  const void* aux;        /* stats_mode O2M_STATS_DOT: tensor of y's shape and dtype.  The epilogue then emits
                             stats[(m / R) * Co * 2 + o * 2] = sum over the R pixels of acc[., o] * aux[., o], acc =
                             the convolution result BEFORE out_scale (0 in the odd slots), i.e. the data gradient of
                             a zero-padded modulated conv leaves the kernel already multiplied by the style
                             (out_scale = s) with its style dot sum_p g * x (aux = the layer's input x) as row-block
                             partials: the separate o2m_fold_scale_dot pass over the gradient and x (layers.py:152-154
                             backward) is not run.  Requires R > 0, act NONE, no bias, no residual.
                             o2m_conv2d_dots_finalize adds a sample's partials in block order. */
  void* aux_scaled;       /* O2M_STATS_DOT only, or NULL: receives aux * out_scale[b, o] (y's shape and dtype) -- the
                             modulated input x * s that o2m_conv2d_wgrad reduces, written while aux is in registers */
} o2m_conv_desc;
#define O2M_STATS_MOMENTS 0
#define O2M_STATS_DOT 1
int o2m_conv2d_fwd(const o2m_conv_desc* d, void* stream);
/* The border ring of the data gradient of a 3 x 3 conv behind ReflectionPad2d(1) (blocks.py:17-26,45-57: every residual
 * block of the generator).  d describes the zero-padded data-gradient conv on the CROPPED domain -- x = the gradient of the
 * conv's output [B][H][W][Ci], w = the flipped filter [Co][3][3][Ci], y = [B][H][W][Co], bf16, pad / pad_mode ignored -- and y
 * must already hold that conv's result (o2m_conv2d_fwd with pad 1, O2M_PAD_ZERO; same stream).  ADDED to y, with packed
 * bf16 atomics: full[oy][ox] at y[R(oy - 1)][R(ox - 1)] for the ring oy in {0, H + 1} or ox in {0, W + 1} of the full
 * correlation on the padded (H + 2) x (W + 2) domain, R(-1) = 1, R(H) = H - 2: the adjoint of the pad applied to what the
 * cropped conv leaves out.  Together the two launches equal o2m_conv2d_fwd on the padded domain (pad 2) + the fold
 * (o2m_conv_desc.fold_pad / o2m_fold_scale_dot), without the 6 % more GEMM rows, the tail launch they cost the
 * 256 x 256-tile kernel, the zero-fill launch and the fold pass over a 66 x 66 map.  Run-to-run differences in the last
 * bit of the ring's targets (atomics): not for the deterministic mode.  Ci % 64 == 0 (or 32), Co % 64 == 0, H, W >= 4. */
int o2m_conv2d_reflect_border(const o2m_conv_desc* d, void* stream);
/* Rows per InstanceNorm partial for this problem (the tile configuration o2m_conv2d_fwd would select),
 * or 0 when the epilogue cannot emit them (Ho*Wo not a multiple of the tile's row block: the odd-sized
 * discriminator maps -- the caller then runs o2m_instnorm_stats).  d->stats itself is not read. */
int32_t o2m_conv2d_stats_rows(const o2m_conv_desc* d);
/* Partial rows PER SAMPLE the epilogue writes to d->stats for this problem: Ho*Wo / o2m_conv2d_stats_rows(d) for the
 * kernels whose partials cover consecutive pixels, 4 x (8 x 32 tiles per sample, clipped at the map's edge) for the
 * halo-tile kernel of the 4 x 4 trunk convolutions on their odd-sized maps (each partial then covers the pixels of its
 * tile's wave row that exist); 0: the epilogue cannot emit them (run o2m_instnorm_stats).  The stats workspace holds
 * B * chunks * Co * 2 floats; o2m_instnorm_finalize takes `chunks` as its nchunks with P = Ho*Wo. */
int32_t o2m_conv2d_stats_chunks(const o2m_conv_desc* d);
/* dots[b][c] = sum over the nchunks = Ho*Wo / R row blocks of sample b of the O2M_STATS_DOT partials
 * (partial [B][nchunks][C][2], even slots), in block order: fixed summation order, no atomics. */
int o2m_conv2d_dots_finalize(const float* partial, float* dots, int32_t B, int32_t C, int32_t nchunks,
                             void* stream);

/* Per-tensor fp8 quantisation (config #5).  Two launches per tensor, no host round trip, no atomics:
 *   o2m_amax         : partial[k] = max |x| over block k's share, k < O2M_AMAX_PARTIALS (all written)
 *   o2m_quantize_fp8 : amax = max_k partial[k];  scale = FMT_MAX / max(amax, 1e-12)  (448 for e4m3, 57344 for
 *                      e5m2);  y[i] = fp8(x[i] * scale)  (round to nearest even, saturating);
 *                      deq[0] = 1 / scale, deq[1] = amax
 * x is `dtype` (O2M_BF16 / O2M_F32), y is `fmt` (O2M_FP8_E4M3 / O2M_BF8_E5M2) bytes, n % 8 == 0. */
#define O2M_AMAX_PARTIALS 1024
int o2m_amax(const void* x, float* amax, int64_t n, int32_t dtype, void* stream);
int o2m_quantize_fp8(const void* x, const float* amax, void* y, float* deq, int64_t n, int32_t dtype,
                     int32_t fmt, void* stream);
/* Delayed scaling: ONE pass.  The scale comes from `amax_prev` -- the O2M_AMAX_PARTIALS partial maxima of the tensor the
 * same call site quantised last time (written by o2m_amax or by this function) -- and the partial maxima of THIS tensor go
 * to `amax_next` (all O2M_AMAX_PARTIALS slots written; a different buffer) for the next call.  Values beyond the previous
 * amax saturate.  deq[0] = 1 / scale, deq[1] = the amax the scale was made from. */
int o2m_quantize_fp8_delayed(const void* x, const float* amax_prev, void* y, float* deq, float* amax_next, int64_t n,
                             int32_t dtype, int32_t fmt, void* stream);

/* Kernel-side forms of one equalised-LR filter (layers.py:12-24: W*c is recomputed on every
 * forward).  w is the parameter, fp32 [Co][Ci][KK] (KK = KH*KW).  Written:
 *   full fp32 [Cop][KK][Cip] = W*c, zero beyond (Co, Ci);   w_f = full in `dtype`;
 *   w_d (`dtype`) [Cip][KK][Cop] = the data-gradient filter (taps reversed, in/out swapped);
 *   q fp32 [Cop][Cip] = sum_kk full^2 and qt = its transpose (layers.py:156-161 factored; both
 *   NULL for an unmodulated conv).
 */
int o2m_prepare_weights(const float* w, float* full, void* w_f, void* w_d, float* q, float* qt,
                        int32_t Co, int32_t Ci, int32_t KK, int32_t Cop, int32_t Cip, float c,
                        int32_t dtype, void* stream);

/* The same for EVERY filter of a network in one launch (54 launches per step otherwise).  `jobs` is a DEVICE array
 * the caller builds once per network (the pointers are stable: parameters live in one flat bucket, the outputs are
 * per-layer buffers reused every step); job j owns blocks [first_block, next job's first_block) of 256 threads,
 * ceil(Cop * Cip / 256) of them; total_blocks = their sum.  `dtype` = element type of every w_f / w_d. */
typedef struct o2m_prep_job {
  const float* w;  /* [Co][Ci][KK] parameter                      */
  float* full;     /* [Cop][KK][Cip] fp32 W*c                     */
  void* w_f;       /* [Cop][KK][Cip] `dtype`                      */
  void* w_d;       /* [Cip][KK][Cop] `dtype`, taps reversed       */
  float* q;        /* [Cop][Cip] or NULL                          */
  float* qt;       /* [Cip][Cop] or NULL (with q)                 */
  int32_t Co, Ci, KK, Cop, Cip;
  float c;
  int32_t first_block;
  int32_t reserved;
} o2m_prep_job;
int o2m_prepare_weights_batched(const o2m_prep_job* jobs, int32_t n_jobs, int32_t total_blocks, int32_t dtype,
                                void* stream);

/* Per-sample pre-modulated filters for the forward modulated conv:
 *   out[b][o][kh][kw][i] = (dtype) ( w32[o][kh][kw][i] * s[b][i] )
 * i.e. layers.py:152-154 (weights * s) without the demodulation, which stays an epilogue
 * scale.  w32 is fp32 [Co][KH*KW][Ci]; rounding to bf16 happens AFTER the style is folded in.
 */
int o2m_modulate_weights(const float* w32, const float* s, void* out, int32_t B, int32_t Co,
                         int32_t KK, int32_t Ci, int32_t dtype, void* stream);

/* Style path of the modulated conv (to_style linear + demodulation, layers.py:138-161) and
 * its backward, B x C sized (see csrc/style.hip for the formulas).
 *   fwd: w [B][WD], Ws [Ci][WD] (raw to_style weight, cs = 1/sqrt(WD) applied inside), bs [Ci],
 *        Qt [Cip][Cop] = transposed Q (NULL with d NULL: no demodulation)
 *        -> s [B][Cip] (zero beyond Ci), d [B][Cop].
 *   bwd: sums [B][2][Cop] from o2m_act_bwd_reduce, bias [Cop] or NULL, dots [B][Cip] from
 *        o2m_fold_scale_dot, Q [Cop][Cip] -> e [B][Cop], gs [B][Cip] (workspaces),
 *        gw [B][WD], gWs [Ci][WD], gbs [Ci]; gq [Cop][Cip] += dL/dQ (accumulated: the layer may be
 *        used several times in one backward).   d NULL: no demodulation.
  * accumulate != 0: gWs / gbs are ADDED to (one writer per element: pass the parameters' .grad
 *   buffers and the five uses of a decoder layer per step need no autograd additions).
 */
int o2m_style_fwd(const float* w, const float* Ws, const float* bs, const float* Qt, float* s,
                  float* d, int32_t B, int32_t WD, int32_t Ci, int32_t Cip, int32_t Cop, float cs,
                  float eps, void* stream);
int o2m_style_bwd(const float* sums, const float* bias, const float* dots, const float* s,
                  const float* d, const float* Q, const float* w, const float* Ws, float* e,
                  float* gs, float* gw, float* gWs, float* gbs, float* gq, int32_t B, int32_t WD,
                  int32_t Ci, int32_t Cip, int32_t Cop, float cs, int32_t accumulate, void* stream);

/* ------------------------------------------------------------------------------------
 * Weight gradient of the convolution above (the wgrad half of aten::convolution_backward
 * for the same call sites), split over the B*Ho*Wo reduction, fp32 atomics into dw:
 *   dw[o,kh,kw,i] += sum_{b,oy,ox} (gy_scale[b,o]*gy[b,oy,ox,o]) *
 *                                   (in_scale[b,i]*xpad[b,oy+kh-pad,ox+kw-pad,i])
 * dw is fp32 [Co][KH][KW][Ci] and must be zeroed (or hold the running sum) by the caller.
 */
typedef struct {
  const void* x;         /* [B][H][W][Ci]      */
  const void* gy;        /* [B][Ho][Wo][Co]    */
  float* dw;             /* [Co][KH][KW][Ci]   */
  const float* in_scale; /* [B][Ci] or NULL    */
  const float* gy_scale; /* [B][Co] or NULL    */
  int32_t B, H, W, Ci, Co, KH, KW, pad, pad_mode, dtype;
  int32_t splits;        /* >=1: number of slices of the pixel reduction; 0 = auto */
  int32_t reserved0;     /* (round 1-2: segment count of a multi-operand launch; removed with the batched decoder groups) */
  int32_t stride;        /* 0/1: unit stride; >1: gy is the output of a strided conv
                            (Ho = (H + 2 pad - KH) / stride + 1)                          */
  int32_t kernel_hint;   /* 0: the register-staged tiles.  O2M_WGRAD_HINT_P8 (1): the phase-pipelined 256 x 256
                            kernel (LDS-DMA fills, transposing LDS reads) where it applies -- Co = Ci = 256, 3 x 3, pad 1,
                            64-pixel rows, bf16, slab mode -- else as 0.  Alone on the chip it is 1.1-1.3x faster; it
                            holds a whole CU per block (128 KB of LDS), so beside another stream's kernels the smaller
                            tiles pack better: the caller chooses. */
  float* slabs;          /* NULL: the pixel slices add into dw with fp32 atomics (arrival order: the last
                            bits differ from run to run).  Otherwise a workspace of
                            o2m_conv2d_wgrad_slab_floats(d) floats: every slice STORES its Co x K partial
                            there and a second kernel adds the slices to dw in slice order -- bitwise
                            reproducible (the reference's deterministic_cuda_kernels switch, train.py:41-45)
                            and faster: plain stores run at ~4-5x the chip-wide float-atomic rate. */
} o2m_wgrad_desc;
#define O2M_WGRAD_HINT_P8 1
int o2m_conv2d_wgrad(const o2m_wgrad_desc* d, void* stream);
/* Floats of slab workspace o2m_conv2d_wgrad needs for this problem (0 for an invalid descriptor). */
size_t o2m_conv2d_wgrad_slab_floats(const o2m_wgrad_desc* d);

/* End-of-backward conversion of an accumulated weight gradient to the parameter layout:
 *   grad[o][i][kh][kw] += c * ( acc[o][kh][kw][i] + 2 * gq[o][i] * w32[o][kh][kw][i] )
 * and acc / gq are cleared.  acc: fp32 [Cop][KK][Cip] that o2m_conv2d_wgrad added into over
 * every use of the layer; gq (NULL for plain convs): dL/dQ from o2m_style_bwd; w32 = W*c in the
 * same layout; grad: the parameter's .grad, [Co][Ci][KH][KW].  Replaces the permute / scale /
 * AccumulateGrad chain of eager autograd (layers.py:19-24: d/dW = c * d/d(W*c)).
 */
int o2m_wgrad_finalize(float* acc, float* gq, const float* w32, float* grad, int32_t Co, int32_t Ci,
                       int32_t KK, int32_t Cop, int32_t Cip, float c, void* stream);
/* The same for EVERY layer a backward pass left pending, in one launch (+ one that clears the dL/dQ tables when
 * any_gq != 0): `jobs` is a DEVICE array; job j owns blocks [first_block_j, first_block_j + o2m_wgrad_finalize_blocks(Cop,
 * KK, Cip)), first_block ascending from 0, total_blocks = their sum.  29 launches per D+G step become 2 + 2. */
typedef struct o2m_wfin_job {
  float* acc;        /* [Cop][KK][Cip] accumulator (cleared)        */
  float* gq;         /* [Cop][Cip] dL/dQ or NULL (cleared)          */
  const float* w32;  /* [Cop][KK][Cip] W*c (read when gq != NULL)   */
  float* grad;       /* [Co][Ci][KK] the parameter's .grad (+=)     */
  int32_t Co, Ci, KK, Cop, Cip;
  float c;
  int32_t first_block;
  int32_t reserved;
} o2m_wfin_job;
int32_t o2m_wgrad_finalize_blocks(int32_t Cop, int32_t KK, int32_t Cip);
int o2m_wgrad_finalize_batched(const o2m_wfin_job* jobs, int32_t n_jobs, int32_t total_blocks, int32_t any_gq, void* stream);

/* ------------------------------------------------------------------------------------
 * Backward of the fused epilogue: gu = g * act'(y), plus the per-(b,c) sums the
 * modulated conv and the bias need:
 *   sums[b,0,c] = sum_p gu ,  sums[b,1,c] = sum_p gu * (y - residual)
 * (second sum gives d loss / d out_scale = sums1 / out_scale for act in {none, relu}).
 * sums is fp32 [B][2][C], zeroed by the caller and ADDED to: with fp32 atomics from every pixel
 * chunk (partials NULL), or -- the reference's deterministic_cuda_kernels mode, train.py:41-45 --
 * through `partials`, caller workspace of o2m_chan_partials_floats(B, P, C, 2) floats: every chunk
 * stores its row there and a second launch adds the rows in chunk order (bitwise reproducible).
 * If out_mul
 * ([B][C] fp32) is given the STORED tensor is gu * out_mul[b,c] (the demodulation factor is
 * folded here so that dgrad and wgrad of the modulated conv read it pre-scaled); the sums
 * always use the unscaled gu.
 * Replaces the ReLU/LeakyReLU/Tanh backward and the bias reduction of
 * convolution_backward.  `y` is the forward OUTPUT (post-activation).
 * Reduce-only form (the bias gradient of a conv with no activation, layers.py:84-100): y = NULL
 * with act = O2M_ACT_NONE and no residual; gu may then be NULL as well (nothing is stored).
 */
size_t o2m_chan_partials_floats(int32_t B, int32_t P, int32_t C, int32_t nv);
int o2m_act_bwd_reduce(const void* g, const void* y, const void* residual,
                       const float* out_mul, void* gu, float* sums, int32_t B, int32_t P,
                       int32_t C, int32_t act, int32_t dtype, float* partials, void* stream);

/* Backward of ReflectionPad2d fused with the style scale and the style-gradient dot:
 *   gfold = fold_reflect(gpad)             (pad == 0: identity)
 *   gx[b,y,x,c]   = gfold[b,y,x,c] * scale[b,c] + gres[b,y,x,c]   (scale NULL: 1; gres NULL: 0)
 *   dots[b,c]    += sum_{y,x} gfold[b,y,x,c] * x[b,y,x,c]   (dots/x NULL: skipped)
 *   xs[b,y,x,c]   = x[b,y,x,c] * scale[b,c]                  (xs NULL: skipped; needs dots)
 * gpad is [B][H+2p][W+2p][C]; gx, x, xs, gres are [B][H][W][C]; dots fp32 [B][C] zeroed by caller.
 * xs is the modulated input, a by-product for o2m_conv2d_wgrad (x is being read anyway).
 * gres: the gradient that reaches the layer's input through a residual connection around it
 * (blocks.py:33,68: x + block(x)) -- added here instead of by a separate elementwise pass.
 * Fused activation backward (act_sums != NULL; the ModulatedResnetBlock's conv -> ReLU -> conv chain,
 * blocks.py:49-57): x is ALSO the output y of the layer below, which applied `act` and the
 * demodulation act_mul[b,c]; the stored tensor is then that layer's o2m_act_bwd_reduce result
 *   gx <- gx * act'(x) * act_mul[b,c],  act_sums[b,0,c] += sum gx act'(x),  act_sums[b,1,c] += sum gx act'(x) x
 * (act_sums fp32 [B][2][C], zeroed by the caller; act in {none, relu, lrelu}).
 * partials: NULL (fp32 atomics) or o2m_chan_partials_floats(B, H*W, C, act_sums ? 3 : 1) floats of
 * workspace for the ordered two-stage sums, as in o2m_act_bwd_reduce.
 */
int o2m_fold_scale_dot(const void* gpad, const void* x, const float* scale, void* gx,
                       float* dots, void* xs, const void* gres, int32_t act, const float* act_mul,
                       float* act_sums, int32_t B, int32_t H, int32_t W, int32_t C,
                       int32_t pad, int32_t dtype, float* partials, void* stream);

/* ------------------------------------------------------------------------------------
 * InstanceNorm2d (eps, biased variance, no affine; builder.py:164,172,273..; blocks.py:23,27)
 * fused with the activation behind it and the residual add of ResnetBlock (blocks.py:33).
 *   stats : partial[b,chunk,c,{sum,sumsq}] over pixel chunks (deterministic two-stage),
 *           then mean_rstd[b,c,{mean,rstd}].
 *   apply : y = act((x-mean)*rstd) + residual
 *   bwd   : with xh=(x-mean)*rstd, gh = g*act'(xh):  gx = rstd*(gh - mean_p(gh) - xh*mean_p(gh*xh))
 * `partial` is caller workspace of o2m_instnorm_ws_floats(B,P,C) floats.
 */
size_t o2m_instnorm_ws_floats(int32_t B, int32_t P, int32_t C);
/* Second stage alone: partial [B][nchunks][C][2] (from o2m_conv2d_fwd's epilogue, nchunks = P / R) ->
 * mean_rstd [B][C][2]. */
int o2m_instnorm_finalize(const float* partial, float* mean_rstd, int32_t B, int32_t P, int32_t C,
                          int32_t nchunks, float eps, void* stream);
int o2m_instnorm_stats(const void* x, float* partial, float* mean_rstd, int32_t B, int32_t P,
                       int32_t C, float eps, int32_t dtype, void* stream);
int o2m_instnorm_apply(const void* x, const float* mean_rstd, const void* residual, void* y,
                       int32_t B, int32_t P, int32_t C, int32_t act, int32_t dtype, void* stream);
int o2m_instnorm_bwd(const void* g, const void* x, const float* mean_rstd, float* partial,
                     float* gsums, void* gx, int32_t B, int32_t P, int32_t C, int32_t act,
                     int32_t dtype, void* stream);

/* InstanceNorm + activation + DownSample in one pass, and its backward (builder.py:170-173,272-282: conv ->
 * InstanceNorm2d -> ReLU / LeakyReLU -> DownSample in the encoder and in the discriminator / style-extractor trunk).
 *   fwd: y = D( act( (x - mean) * rstd ) ), D = the banded operator (sy, wy, sx, wx) of o2m_resample2d below with T = 4
 *        taps per axis and starts 2 or 3 apart (DownSample on even / odd sizes); O2M_ERR_UNSUPPORTED for any other
 *        operator (the caller runs o2m_instnorm_apply + o2m_resample2d).  The normalised map is never stored.
 *   bwd: gx = InstanceNorm_backward( D^T g_coarse ) with the taps of the TRANSPOSED operator (T = 2 per axis): the fine
 *        gradient D^T g is gathered per pixel in both passes (sums, apply) instead of being written once and read twice.
 *        partial / gsums as in o2m_instnorm_bwd.  x [B][H][W][C], g_coarse [B][Hl][Wl][C].
 */
int o2m_instnorm_act_resample2d(const void* x, const float* mean_rstd, void* y, const int32_t* sy, const float* wy,
                                const int32_t* sx, const float* wx, int32_t B, int32_t H, int32_t W, int32_t Ho,
                                int32_t Wo, int32_t C, int32_t T, int32_t span_y, int32_t span_x, int32_t act,
                                int32_t dtype, void* stream);
int o2m_instnorm_resample_bwd(const void* g_coarse, const void* x, const float* mean_rstd, float* partial,
                              float* gsums, void* gx, const int32_t* sy, const float* wy, const int32_t* sx,
                              const float* wx, int32_t B, int32_t H, int32_t W, int32_t Hl, int32_t Wl, int32_t C,
                              int32_t T, int32_t act, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * Separable banded resampling: y[b,oy,ox,c] = sum_{ty,tx} wy[oy,ty]*wx[ox,tx] *
 *                                             x[b, sy[oy]+ty, sx[ox]+tx, c]
 * One kernel serves Smooth (layers.py:207-214), UpSample = bilinear x2 then blur
 * (layers.py:223-229), DownSample = blur then bilinear to floor(H/2) (layers.py:241-247,
 * fractional taps for odd sizes) and the transposes of all three (their backward); the
 * host composes the 1-D operators (one_to_many_gan_amd/resample.py) and passes the taps.
 * sy/sx: int32 [Ho]/[Wo] first source index; wy/wx: fp32 [Ho][T]/[Wo][T], zero padded.
 * Taps must stay in range: 0 <= s[o] and s[o]+T <= source size (host guarantees).
 * Ty / Tx: taps per axis (wy is [Ho][Ty], wx is [Wo][Tx]).  span_y / span_x: if that axis' starts
 * are non-decreasing, its largest step s[o+1]-s[o] (>= 1: selects the kernel that shares one
 * (Ty+span_y) x (Tx+span_x) input patch between 2x2 outputs); 0 = no promise, one output per
 * thread (needs Ty == Tx).  A 6-tap operator (the transposed upsample) is run by the host as a
 * vertical and a horizontal pass with identity taps on the other axis.
 */
int o2m_resample2d(const void* x, void* y, const int32_t* sy, const float* wy,
                   const int32_t* sx, const float* wx, int32_t B, int32_t H, int32_t W,
                   int32_t Ho, int32_t Wo, int32_t C, int32_t Ty, int32_t Tx, int32_t span_y,
                   int32_t span_x, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * Device-resident input pipeline: the reference keeps every image in
 * host RAM as a normalised float tensor (datasets.py:34-42), flips it per fetch
 * (datasets.py:44-50) and ships batches through 3 x 8 DataLoader workers + pinned H2D copies
 * (train.py:128-165).  Here the images stay in HBM as uint8 NHWC; one launch builds a batch:
 *   out[b][y][x][c] = ((pool[index[b]][y][flip[b] ? W-1-x : x][c] / 255) - 0.5) / 0.5
 * = ToTensor + Normalize((0.5,), (0.5,)) (train.py:120-126) + RandomHorizontalFlip, evaluated
 * in fp32 in that order, then stored as `dtype` with channels C..Cp-1 zeroed.
 * pool: uint8 [N][H][W][C]; index: int32 [B] (no range check on the device: the host
 * guarantees 0 <= index[b] < N); flip: uint8 [B]; out: `dtype` [B][H][W][Cp].
 */
int o2m_gather_images(const uint8_t* pool, const int32_t* index, const uint8_t* flip, void* out,
                      int32_t N, int32_t B, int32_t H, int32_t W, int32_t C, int32_t Cp,
                      int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * Adaptive discriminator augmentation (reference call sites train.py:175-188,
 * training.py:100,104,200; the transforms live in the un-vendored dependency pytorch-ada, so
 * these follow the published StyleGAN2-ADA pipe -- see one_to_many_gan_amd/ada.py, oracle/ada.py;
 * parity unpinned).  Images are NHWC with Cp (multiple of 8) stored channels, C (1 or 3) real.
 *
 * o2m_ada_grid_sample: F.affine_grid(theta, align_corners=False) + F.grid_sample(bilinear,
 *   zeros padding, align_corners=False) of x [B][Hs][Ws][Cp] into y [B][Ho][Wo][Cp];
 *   theta fp32 [B][2][3] in normalised coordinates.
 * o2m_ada_grid_sample_bwd: its adjoint, as a gather (one thread per source pixel enumerates the
 *   outputs whose bilinear footprint covers it; no atomics); gx [B][Hs][Ws][Cp] in `dtype`, fully
 *   written (zeros outside the sampled region).
 * o2m_reflect_fold: adjoint of F.pad(mode="reflect") with margins (pad_left, Wp-W-pad_left,
 *   pad_top, Hp-H-pad_top), each smaller than the image; gpad [B][Hp][Wp][Cp] (in_dtype) ->
 *   gx [B][H][W][Cp] (out_dtype).
 * o2m_ada_colour: y[c] = sum_k m[b][c][k] x[k] + m[b][c][3] on the C real channels
 *   (m fp32 [B][3][4]; C = 1 uses m[b][0][0] and m[b][0][3]); padding channels written as 0.
 *   P = pixels per sample.  Its adjoint is the same call with the transposed matrix and zero
 *   offsets.
 */
int o2m_ada_grid_sample(const void* x, const float* theta, void* y, int32_t B, int32_t Hs,
                        int32_t Ws, int32_t Ho, int32_t Wo, int32_t Cp, int32_t dtype, void* stream);
int o2m_ada_grid_sample_bwd(const void* gy, const float* theta, void* gx, int32_t B, int32_t Hs,
                            int32_t Ws, int32_t Ho, int32_t Wo, int32_t Cp, int32_t dtype,
                            void* stream);
int o2m_reflect_fold(const void* gpad, void* gx, int32_t B, int32_t H, int32_t W, int32_t Hp,
                     int32_t Wp, int32_t pad_top, int32_t pad_left, int32_t Cp, int32_t in_dtype,
                     int32_t out_dtype, void* stream);
int o2m_ada_colour(const void* x, const float* m, void* y, int32_t B, int64_t P, int32_t C,
                   int32_t Cp, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * Layout conversion at the public (logical NCHW fp32) boundary.
 *   pack  : NCHW fp32 [B][C][H][W]  -> NHWC `dtype` [B][H][W][Cp] (channels >= C zeroed)
 *   unpack: NHWC `dtype` [B][H][W][Cp] -> NCHW fp32 [B][C][H][W]
 */
int o2m_pack_nchw(const float* src, void* dst, int32_t B, int32_t C, int32_t H, int32_t W,
                  int32_t Cp, int32_t dtype, void* stream);
int o2m_unpack_nhwc(const void* src, float* dst, int32_t B, int32_t C, int32_t H, int32_t W,
                    int32_t Cp, int32_t dtype, void* stream);

/* Adversarial (LSGAN) loss of the discriminator's patch map and the reference's confidence (training.py:111-118,202:
 * ((D(real) - 1)^2).mean(), (D(fake)^2).mean(), sign(2 D(x) - 1).mean()) in ONE launch; replaces F.mse_loss + the
 * sign / mean chain (eleven elementwise launches per discriminator step).  scores: internal [N][P][C] (`dtype`), channel
 * 0 = the logical score.  out (4 floats): out[0] = sum over samples [0, n_first) of (s - t0)^2, out[1] = the same over
 * [n_first, N) with t1, out[2] / out[3] = sum of sign(2 s - 1) over the two halves (fixed summation order).
 * o2m_lsgan_bwd: g_scores[n][p][0] = coef[half] * 2 * (s - t_half) with coef a DEVICE pointer to two floats (the upstream
 * gradients of out[0], out[1]); channels 1.. of g_scores are zeroed. */
int o2m_lsgan_fwd(const void* scores, float* out, int32_t N, int32_t P, int32_t C, int32_t n_first, float t0, float t1,
                  int32_t dtype, void* stream);
int o2m_lsgan_bwd(const void* scores, const float* coef, void* g_scores, int32_t N, int32_t P, int32_t C, int32_t n_first,
                  float t0, float t1, int32_t dtype, void* stream);
/* ------------------------------------------------------------------------------------
 * Loss reductions (F.l1_loss / F.mse_loss at training.py:111-112,178,188,202;
 * kl_loss_func loss.py:82-92; path_loss_func loss.py:98-111).  All write per-block fp32
 * partial sums (deterministic; the host adds them) -- no host sync, graph-capturable.
 *   mode O2M_RED_L1   : sum |a-b|
 *   mode O2M_RED_SQ   : sum w[b]*(a-b)^2        (b NULL: 0; w NULL: 1)
 *   mode O2M_RED_MOM  : {sum a, sum a^2}        (two partials per block)
 * and their elementwise backward:
 *   O2M_RED_L1 : ga = coef[0]*sign(a-b)
 *   O2M_RED_SQ : ga = coef[0]*w[b]*(a-b)
 *   O2M_RED_MOM: ga = coef[0] + coef[1]*a
 * coef is a DEVICE pointer to fp32 scalars (upstream gradient folded in by the host ops).
 * n_per_sample = elements per batch sample (H*W*C); partials has o2m_reduce_blocks(n)
 * entries per output.
 */
#define O2M_RED_L1 0
#define O2M_RED_SQ 1
#define O2M_RED_MOM 2
int32_t o2m_reduce_blocks(int64_t n);
int o2m_reduce_fwd(const void* a, const void* b, const float* w, float* partials, int32_t B,
                   int64_t n_per_sample, int32_t mode, int32_t dtype, void* stream);
int o2m_reduce_bwd(const void* a, const void* b, const float* w, const float* coef, void* ga,
                   int32_t B, int64_t n_per_sample, int32_t mode, int32_t dtype, void* stream);
/* Backward of a pair term  sum_b w[b] * sum (a[b] - b[b])^2  whose operands ALSO feed other consumers (path_loss_func on
 * the decoder features, loss.py:98-111: every feature map goes on to the next decoder layer): in one pass
 *   ga = gin_a + coef[0] * w[b] * (a - b) ,   gb = gin_b - coef[0] * w[b] * (a - b)
 * with gin_a / gin_b = the gradient arriving from the other consumer (NULL: zero).  Replaces reduce_bwd + a negation
 * pass + autograd's accumulation add (11 -> 6 passes over a half-batch feature map). */
int o2m_pair_grad(const void* a, const void* b, const float* w, const float* coef, const void* gin_a, const void* gin_b,
                  void* ga, void* gb, int32_t B, int64_t n_per_sample, int32_t dtype, void* stream);

/* ------------------------------------------------------------------------------------
 * Fused Adam over one flat fp32 bucket (torch.optim.Adam at train.py:94-116: no weight
 * decay, eps 1e-8).  `step` is a DEVICE fp32 scalar holding the 1-based step count (the
 * host bumps it with a device-side add, so the update is graph-capturable).
 *   m = b1*m + (1-b1)*g ; v = b2*v + (1-b2)*g*g
 *   p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
 * grad_scale multiplies g first (1/world_size after a sum all-reduce).
 */
int o2m_adam_step(float* p, const float* g, float* m, float* v, const float* step, int64_t n,
                  float lr, float beta1, float beta2, float eps, float grad_scale, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* O2M_HIP_H */

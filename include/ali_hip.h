/*
 * ali_hip.h -- C ABI of libali_hip.so: the MI355X (gfx950) kernels behind the
 * ALI/BiGAN training path of wtaylor17/ImageCFGen-Pytorch (image_scms package).
 *
 * The reference has no FFI of its own: all of its arithmetic is dispatched by
 * torch.nn modules.  Each entry point below names the reference call site
 * (file:line, relative to the reference checkout) whose ATen op it replaces.
 *
 * Conventions
 *   - plain pointers + sizes, no torch types; all tensors fp32 in HBM.
 *   - activations are NHWC ([B,H,W,C], C contiguous); an op that reads an
 *     activation with float4 loads needs its channel stride % 4 == 0 (the
 *     callers pad 5->8 and 771->772 channels with zeros).
 *   - every function is stream ordered, never allocates, never synchronises,
 *     and returns 0 or a negative AliStatus; ali_last_error() gives the text.
 *   - `ws` is caller-provided scratch (>= the matching *_workspace_bytes());
 *     it may be reused by the next call on the same stream, not by concurrent
 *     streams.  Its first ALI_WS_RESERVED bytes are the split-K arrival
 *     counters of the GEMM kernels: zero-fill a new workspace once
 *     (hipMemset); every launch leaves them at zero again.
 */
#ifndef ALI_HIP_H
#define ALI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* ali_stream_t; /* hipStream_t */
#define ALI_WS_RESERVED 4096

typedef enum {
  ALI_OK = 0,
  ALI_ERR_BAD_ARG = -1,
  ALI_ERR_WORKSPACE = -2,
  ALI_ERR_LAUNCH = -3
} AliStatus;

typedef enum { ALI_ACT_NONE = 0, ALI_ACT_LEAKY = 1, ALI_ACT_TANH = 2 } AliAct;

/* Geometry of one Conv2d (cross-correlation) y = conv(x, W):
 *   x [B,H,W,C] (channel stride C, may include zero padding channels)
 *   y [B,P,Q,K] (channel stride K)
 *   W [K][C][R][S] logically; square stride / padding.
 * A ConvTranspose2d (mnist.py:64-72, audio_mnist.py:216-242) is described by
 * the Conv2d it is the data-gradient of: x := its output, y := its input. */
typedef struct {
  int32_t B, H, W, C;
  int32_t P, Q, K;
  int32_t R, S, stride, pad;
} AliConvGeom;

/* Optional fused epilogue, applied in this order to v = acc:
 *   v += bias[n]; v = act(v); v *= mask[img*mask_ld + n]; v *= act'(dact_y)
 * dact_y has the layout of the output; act' is evaluated on the *activated*
 * value saved in forward (leaky: y>0 ? 1 : slope, tanh: 1-y*y).
 *
 * Fused BatchNorm2d reductions (mnist.py:111,114,118,122; bn_mode != 0): the launch also leaves, per M-tile of its
 * grid ("slot"; ali_conv_mtiles tells how many), the column sums over the tile's rows that nn.BatchNorm2d needs, so no
 * separate pass over the tensor is made:
 *   bn_mode 1 (the conv in front of a BatchNorm, forward): s0 = sum v~, s1 = sum v~^2 with v~ = v * bn_stat_mask[img,n]
 *             (a Dropout2d between the conv and the BatchNorm; NULL: v~ = v, the stored value);
 *   bn_mode 2 (the data-gradient GEMM behind a BatchNorm, backward): s0 = sum g~ * xhat, s1 = sum g~ with
 *             g~ = v * bn_mask_pre[img,n], xhat = (bn_x * bn_mask_in[img,n] - bn_mean[n]) * bn_invstd[n]; bn_x has the
 *             layout of the output.
 * bn_part[(s*N + n) * slots + slot(tile)], slots = number of M-tiles.  bn_groups > 1 (mode 1): the batch is that many
 * passes back to back (B/groups images each, B/groups a multiple of the tile height: ali_conv_mtiles reports it);
 * the slots of pass g are the contiguous range [g*slots/groups, (g+1)*slots/groups).  Consumed by
 * ali_bn_stats_from_partials / ali_bn_bwd_from_partials. */
typedef struct {
  const float* bias;   /* [N] or NULL */
  int32_t act;         /* AliAct */
  float slope;
  const float* mask;   /* per (image, channel) Dropout2d mask or NULL */
  int32_t mask_ld;
  const float* dact_y; /* or NULL */
  int32_t dact;        /* AliAct of the producer of dact_y */
  float dslope;
  float* bn_part;      /* or NULL */
  int32_t bn_mode, bn_groups;
  const float* bn_stat_mask;   /* mode 1, or NULL; row stride bn_mask_ld */
  int32_t bn_mask_ld;
  const float* bn_x;           /* mode 2 */
  const float* bn_mean;
  const float* bn_invstd;
  const float* bn_mask_in;     /* mode 2, or NULL; row stride bn_mask_ld */
  const float* bn_mask_pre;    /* mode 2, or NULL; row stride bn_mask_ld */
  /* Arithmetic of the contraction: 0 = fp32 operands on v_mfma_f32_32x32x2_f32 (exact fp32 fma chain); 1 = operands
   * rounded to fp16 on their way into LDS, v_mfma_f32_32x32x16_f16 with fp32 accumulation (BASELINE config 5,
   * esrf_acoustic.py:134-260) where the layer takes the uniform-tap path (input channel stride % 32 == 0, more
   * than 32 output channels); other layers keep fp32.  Tensors in HBM are fp32 either way. */
  int32_t mfma_f16;
  /* fp16 twins (mfma_f16 launches only; all optional): in16 / w16 = the gathered operand and the packed weights as
   * fp16 arrays of the same shapes -- when both are given and the input channel stride % 64 == 0 the kernel reads them
   * instead of x / w (half the bytes, no conversion); out16 = where to leave the fp16 twin of the output for the next
   * layer.  A launch that cannot run on fp16 MFMA (ali_conv_uses_f16 = 0) ignores in16 / w16 and keeps fp32
   * arithmetic, but still leaves out16, so that the layer behind it reads fp16 operands (ali_conv_writes_out16). */
  const void* in16;
  const void* w16;
  void* out16;
  /* Dispatch order of the launch's M-tiles (device int32[tile_order_n], from ali_conv_tile_order; NULL = natural
   * order).  Ignored unless tile_order_n equals the launch's M-tile count and every block owns a whole k-loop. */
  const int32_t* tile_order;
  int32_t tile_order_n;
  /* Operands that are column ranges of wider row-major buffers (the Discriminator's joint [dx | dz] rows,
   * mnist.py:152-154): in_ld = floats between consecutive pixels of the gathered operand (its fp16 twin alike),
   * out_ld = the same for the output (and dact_y, out16).  0 = dense.  Not with the fused BatchNorm modes. */
  int32_t in_ld;
  int32_t out_ld;
  /* Forward convs: the number of leading input channels that carry data when the channel stride is padded (a first
   * layer's 5 planes in an 8-channel NHWC tensor, mnist.py:108); the others must be zero in x or in w_kxc.  A hint:
   * 0 = unknown; kernels that can skip the padding do. */
  int32_t in_ch_live;
  /* Capacity check of bn_part: the number of slots the caller sized it for (what ali_conv_mtiles told it); a launch
   * whose M-tile count differs (tuning reloaded in between, other precision) fails with ALI_ERR_BAD_ARG instead of
   * writing out of bounds.  0 = unchecked. */
  int32_t bn_slots;
  /* mfma_f16 launches: the fp16 twin of dact_y (same layout), or NULL.  When given, act' is evaluated on it instead of
   * on dact_y -- the sign / value of the previous layer's output as the next layer's GEMM saw it -- and the launch reads
   * 2 instead of 4 bytes per output element (the data gradients of the first layers are bound by exactly these bytes).
   * dact_y must still be passed (it selects the epilogue). */
  const void* dact_y16;
} AliEpilogue;

/* A weight-gradient launch that splits the pixel range writes S partial results ("slabs") into its workspace and
 * sums them in a second launch.  With `fold` given, ali_conv_bwd_weight skips that second launch and describes it here
 * instead (S > 0; S == 0: nothing was deferred, dst is final): the caller gives every such launch of a backward pass its
 * own workspace region (ws_used bytes from the start of the ws it passed stay live) and folds them all with ONE
 * ali_wgrad_fold_multi in front of the optimiser step. */
typedef struct AliWgradFold {
  const float* ws;
  float* dst;
  const float* dbws;
  float* db;
  int64_t slab, s_dc, s_gc, s_tap;
  int32_t S, Mtot, Cg, Cd, Cg_log, Cd_log, T, reserved;
  uint64_t ws_used;
} AliWgradFold;

/* With `job` given (together with `fold`), ali_conv_bwd_weight does not launch its GEMM either when the launch is of
 * the common kind (fp32, 64x64 tiles, pixel table): it records it here (opaque[0] = 1; 0 = it was launched as usual) and
 * the caller issues all recorded launches of a backward pass with ONE ali_wgrad_launch_multi -- in front of
 * ali_wgrad_fold_multi, after the last of their operands has been produced. */
typedef struct AliWgradJob {
  uint64_t opaque[40];
} AliWgradJob;

/* ---- implicit-GEMM convolutions (fp32 MFMA v_mfma_f32_32x32x2_f32) -------
 * ali_conv_fwd       : nn.Conv2d forward  (mnist.py:31-39,100-135; c2d(...) in
 *                      audio_mnist.py:186-198 etc.) and ConvTranspose2d dgrad.
 *                      w_kxc = packed [K][R*S][C]  (ali_pack_weights).
 * ali_conv_bwd_data  : Conv2d data gradient and nn.ConvTranspose2d forward
 *                      (mnist.py:64-72; ct2d(...) audio_mnist.py:228-242) and
 *                      nn.Linear+Unflatten (audio_mnist.py:226-227) as a 4x4
 *                      transposed conv on a 1x1 map. stride 2 is decomposed
 *                      into sub-pixel phases (no zero insertion).
 *                      w_cxk = packed [C][R*S][K].
 * ali_conv_bwd_weight: weight gradient of either; dst[dc*s_dc+gc*s_gc+t*s_tap]
 *                      with gc = channel of x (logical count Cg_log), dc =
 *                      channel of dy (logical count Cd_log), t = r*S+s.     */
size_t ali_conv_workspace_bytes(const AliConvGeom* g, int32_t which /*0 fwd,1 bwd_data,2 bwd_weight*/);
/* Number of M-tiles (= bn_part slots) ali_conv_fwd (which = 0) / ali_conv_bwd_data (which = 1) will launch for this
 * geometry with AliEpilogue.mfma_f16 = mfma_f16, and the tile height in *tile_rows; 0 for a bad geometry.  Rows are ordered (pixel, image) when
 * *pixel_major != 0 (then bn_groups needs B/groups % tile_rows == 0), (image, pixel) otherwise (then it needs
 * B/groups * P*Q % tile_rows == 0). */
int32_t ali_conv_mtiles(const AliConvGeom* g, int32_t which, int32_t mfma_f16, int32_t* tile_rows,
                        int32_t* pixel_major);
/* Host-side: the M-tile ids of the launch ali_conv_fwd (which = 0) / ali_conv_bwd_data (which = 1) makes for g, sorted
 * by k-loop length, longest first (edge tiles of padded / strided-transposed convolutions skip the taps that fall
 * outside the input).  The workgroup dispatcher deals blocks round-robin over the CUs, so dispatching in this order
 * gives every CU the same mix of long and short tiles.  Writes at most cap ids to order (host memory) and returns their
 * number; 0 when all tiles cost the same or the launch cannot use an order.  Upload once per geometry and pass as
 * AliEpilogue.tile_order. */
int32_t ali_conv_tile_order(const AliConvGeom* g, int32_t which, int32_t mfma_f16, int32_t* order, int32_t cap);
/* 1 if the launch ali_conv_fwd (which = 0) / ali_conv_bwd_data (which = 1) makes for g with mfma_f16 = 1 writes
 * AliEpilogue.out16 (every valid geometry does). */
int32_t ali_conv_writes_out16(const AliConvGeom* g, int32_t which);
/* 1 if that launch, given ep (mfma_f16 = 1), multiplies fp16-rounded operands: the uniform-tap GEMM loop (gathered
 * channel count % 32 == 0) and the first Conv2d of the spectrogram stacks (4 / 8 -> 64 channels, 5x5, stride 2,
 * audio_mnist.py:186); every other launch keeps fp32 arithmetic.  What a checker has to know to emulate the launch. */
int32_t ali_conv_uses_f16(const AliConvGeom* g, int32_t which, const AliEpilogue* ep);
int ali_conv_fwd(const AliConvGeom* g, const float* x, const float* w_kxc, float* y,
                 const AliEpilogue* ep, void* ws, size_t ws_bytes, ali_stream_t stream);
int ali_conv_bwd_data(const AliConvGeom* g, const float* dy, const float* w_cxk, float* dx,
                      const AliEpilogue* ep, void* ws, size_t ws_bytes, ali_stream_t stream);
int ali_conv_bwd_weight(const AliConvGeom* g, const float* x, const float* dy, float* dst,
                        int32_t Cg_log, int32_t Cd_log, int64_t s_dc, int64_t s_gc, int64_t s_tap,
                        float* db /* optional: db[dc] = sum over pixels of dy (Conv2d bias gradient) */,
                        const int32_t* pixtab /* optional: ali_wgrad_pixtab of the same geometry */,
                        int32_t mfma_f16 /* as AliEpilogue.mfma_f16 (needs pixtab); accumulation and slabs stay fp32 */,
                        const void* x16, const void* dy16 /* optional fp16 twins of x / dy (AliEpilogue.out16 of the
                                                             launches that produced them): read instead of x / dy */,
                        int32_t dy_ld /* floats between consecutive pixels of dy; 0 = K (dense).  > K: dy is a column
                                         range of wider rows (AliEpilogue.in_ld / out_ld); twins are then not read */,
                        AliWgradFold* fold /* optional: defer the slab reduction, see AliWgradFold */,
                        AliWgradJob* job /* optional (needs fold): defer the launch itself, see AliWgradJob */,
                        int32_t split_target /* with job: blocks this GEMM should have inside the combined launch
                                                (clamped to 256..1024; 0 = 1024, the stand-alone rule) */,
                        void* ws, size_t ws_bytes, ali_stream_t stream);
/* Independent forward / data-gradient GEMMs in ONE launch.  ali_conv_fwd_job / ali_conv_bwd_data_job take the arguments
 * of ali_conv_fwd / ali_conv_bwd_data and, when the launch is of a kind a multi-job kernel exists for (fp32, uniform-tap
 * loop, 64x64 or 128x32 tiles), record it in *job (opaque[0] = 1) instead of launching; any other launch is issued on
 * `stream` right away (opaque[0] = 0), exactly as the plain entry point would.  ali_gemm_launch_multi then issues the
 * recorded jobs: up to 4 of the same kernel variant per launch, longest k-loops first.  The jobs of one call must be
 * mutually independent (no job reads what another writes), each must have been given a workspace of its own (split-K
 * slabs and arrival counters live there), and every operand named at record time must still be alive and unchanged.
 * mnist.py:224-248: E(x) and G(z), the two branches of the E+G backward pass, D.dx and D.dz are such chains. */
typedef struct AliGemmJob {
  uint64_t opaque[112];
} AliGemmJob;
int ali_conv_fwd_job(const AliConvGeom* g, const float* x, const float* w_kxc, float* y, const AliEpilogue* ep,
                     void* ws, size_t ws_bytes, AliGemmJob* job, ali_stream_t stream);
int ali_conv_bwd_data_job(const AliConvGeom* g, const float* dy, const float* w_cxk, float* dx, const AliEpilogue* ep,
                          void* ws, size_t ws_bytes, AliGemmJob* job, ali_stream_t stream);
int ali_gemm_launch_multi(int32_t n, const AliGemmJob* jobs, ali_stream_t stream);

/* Launches deferred weight-gradient GEMMs (jobs[i].opaque[0] == 1 each) together, 12 per launch, longest blocks first.
 * Every operand and workspace region named at ali_conv_bwd_weight time must still be alive and unchanged. */
int ali_wgrad_launch_multi(int32_t n, const AliWgradJob* jobs, ali_stream_t stream);
/* 1 if ali_conv_bwd_weight would defer the launch for this geometry when given a job (pixel table present): lets the
 * caller count a pass's jobs -- and so choose split_target -- before the first launch. */
int32_t ali_wgrad_deferrable(const AliConvGeom* g, int32_t mfma_f16);
/* Folds the slabs of up to any number of deferred weight-gradient launches (jobs[i].S > 0 each) in as few launches as
 * possible (12 jobs per launch), writing every dst / db. */
int ali_wgrad_fold_multi(int32_t n, const AliWgradFold* jobs, ali_stream_t stream);
/* Per-geometry table for ali_conv_bwd_weight: entry i (2 x int32) of output pixel i = (b,p,q) holds the byte offset of
 * x[b, p*stride, q*stride, 0] and the packed pair (p*stride, q*stride); the kernel adds its tap's (r-pad, s-pad).  With it the kernel's gather
 * addresses cost a table read instead of two integer divisions per 16 bytes (fp32 MFMA shares the vector issue
 * path: that arithmetic is 12-27 % of the kernel's issue slots).  Depends on g only; build once, keep, pass along.
 * `out` holds 2 * B*P*Q int32. */
int ali_wgrad_pixtab(const AliConvGeom* g, int32_t* out, ali_stream_t stream);

/* ---- direct kernels for the one-channel ends of the stacks (VALU + LDS, HBM bound) ----------
 * Stride-1 correlations between a K-channel NHWC map `big` [B,P,Q,K] and a 1-channel map `small`
 * [B,H,W] (H = P+R-1-2*pad), used for ConvTranspose2d(64->1,k4)+Tanh (mnist.py:72-73) and for the single
 * consumed input plane / per-input-channel weight gradient of a first Conv2d (mnist.py:108):
 *   fwd   : out[b,h,w]    = act(bias + sum_{r,s,k} big[b,h+pad-r,w+pad-s,k] * w_tk[r*S+s][k])
 *   dgrad : gbig[b,p,q,k] = act'(y[b,p,q,k]) * sum_{r,s} small[b,p+r-pad,q+s-pad] * w_tk[r*S+s][k]
 *   wgrad : dw[c*s_c + k*s_k + (r*S+s)*s_tap] = sum_{b,p,q} big[b,p,q,k] * small[b,p+r-pad,q+s-pad][c]
 *           for the first nc (<= 8) channels c of `small` (element stride sstride between pixels)
 * `sstride` / `ostride`: element stride of the 1-channel map when it is one plane of an NHWC tensor. */
int ali_tconv1_fwd(const float* big, const float* w_tk, const float* bias, float* out, int32_t B, int32_t P,
                   int32_t Q, int32_t K, int32_t R, int32_t S, int32_t pad, int32_t ostride, int32_t act,
                   float slope, const float* rowscale /* optional: out[b,..] *= rowscale[b * rowscale_ld] (the column of
                   a Dropout2d mask when `out` is one plane of a masked input's gradient) */, int32_t rowscale_ld,
                   ali_stream_t stream);
int ali_tconv1_dgrad(const float* small, int32_t sstride, const float* w_tk, const float* dact_y, int32_t dact,
                     float dslope, float* gbig, int32_t B, int32_t P, int32_t Q, int32_t K, int32_t R, int32_t S,
                     int32_t pad, ali_stream_t stream);
int ali_tconv1_wgrad(const float* big, const float* small, int32_t sstride, int32_t nc, float* dw, int64_t s_k,
                     int64_t s_tap, int64_t s_c, int32_t B, int32_t P, int32_t Q, int32_t K, int32_t R, int32_t S,
                     int32_t pad, void* ws, size_t ws_bytes, ali_stream_t stream);

/* dst[(n*T+t)*Cpad + c] = c < C ? src[n*s_n + t*s_tap + c*s_c] : 0.
 * Re-lays reference-layout parameters ([Cout,Cin,kh,kw] Conv2d, [Cin,Cout,kh,kw]
 * ConvTranspose2d, [out,in] Linear) into the kernel layouts.   */
int ali_pack_weights(const float* src, float* dst, int32_t N, int32_t T, int32_t C, int32_t Cpad,
                     int64_t s_n, int64_t s_tap, int64_t s_c, ali_stream_t stream);

/* the same for up to 40 parameters in one launch (all packs of a parameter group after its Adam step):
 * dims[4*i..] = {N, T, C, Cpad}, strides[3*i..] = {s_n, s_tap, s_c} of job i (host arrays). */
int ali_pack_weights_multi(int32_t n_jobs, const float* const* src, float* const* dst,
                           void* const* dst16 /* optional (array and entries): fp16 twin of dst[i], written alongside */,
                           const int32_t* dims, const int64_t* strides, ali_stream_t stream);

/* ---- pointwise / reduction kernels (HBM bound) --------------------------- */
/* gpre = gy * act'(y)  (backward of nn.LeakyReLU / nn.Tanh, mnist.py:32-73) */
int ali_act_bwd(const float* gy, const float* y, float* gpre, int64_t n, int32_t act, float slope,
                ali_stream_t stream);
/* out[c] = sum_rows x[row*ld + c]  (bias gradients). ws >= 2048*C floats.  */
int ali_colsum(const float* x, int64_t rows, int32_t C, int32_t ld, float* out, void* ws, size_t ws_bytes,
               ali_stream_t stream);
/* out = x * mask[img, c]   (nn.Dropout2d forward and backward, mnist.py:99-134) */
int ali_rowmask_mul(const float* x, const float* mask, float* out, int32_t B, int32_t rows_per_img, int32_t C,
                    ali_stream_t stream);
/* counter-based Bernoulli(1-p)/(1-p) masks for production runs: draw i uses the key
 * (seed, *dev_counter, offset + i).  dev_counter (device int64, may be NULL) lets a captured
 * HIP graph produce fresh masks on every replay (the stepper bumps it once per iteration). */
int ali_dropout_mask(uint64_t seed, uint64_t offset, const int64_t* dev_counter, float p, float* out, int64_t n,
                     ali_stream_t stream);

/* All masks of one iteration in one launch: segment j is a [rows][seg_cpad[j]] mask stored at elements
 * [seg_end[j-1], seg_end[j]) with drop probability seg_p[j] (host arrays, n_seg <= 64).  Its first seg_clog[j] columns
 * take, row-major, the draws ali_dropout_mask gives with offset = number of draws of the earlier segments; the
 * remaining (channel padding) columns are 1. */
int ali_dropout_mask_multi(uint64_t seed, const int64_t* dev_counter, const int64_t* seg_end, const float* seg_p,
                           const int32_t* seg_clog, const int32_t* seg_cpad, int32_t n_seg, float* out,
                           ali_stream_t stream);

/* nn.BatchNorm2d in training / eval mode (mnist.py:111,114,118,122).
 * stats: per-channel batch mean / biased var of (mask ? x*mask : x), running
 * stats updated with `momentum` and the unbiased variance when `training`;
 * writes sc = gamma*invstd, sh = beta - mean*sc and mean/invstd for backward.
 * apply: out = (x*sc + sh) * (mask_post ? mask_post[img,c] : 1).
 * bwd_reduce: dgamma = sum g~ * xhat, dbeta = sum g~, g~ = g*(mask_pre?mask_pre:1),
 *   xhat computed from x~ = x*(mask_in?mask_in:1).
 * bwd_apply: gx = gamma*invstd*(g~ - dbeta/N - xhat*dgamma/N)   [batch stats]
 *            gx = gamma*invstd*g~                               [eval]
 *            then gx *= mask_in (if given) and *= leaky'(x) (slope >= 0 given). */
/* groups > 1 (stats, apply): the batch holds that many independent forward passes back to back (B/groups images each);
 * statistics, running-stat updates (in group order) and scale/shift are per pass; group g's mean/invstd/sc/sh live at
 * [g*stat_stride + c]. */
int ali_bn_stats(const float* x, const float* mask, int32_t B, int32_t rows_per_img, int32_t C,
                 const float* gamma, const float* beta, float* running_mean, float* running_var,
                 float momentum, float eps, int32_t training,
                 float* mean, float* invstd, float* sc, float* sh, int32_t groups, int64_t stat_stride,
                 void* ws, size_t ws_bytes, ali_stream_t stream);
int ali_bn_apply(const float* x, const float* sc, const float* sh, const float* mask_in, const float* mask_post,
                 float* out, int32_t B, int32_t rows_per_img, int32_t C, int32_t groups, int64_t stat_stride,
                 ali_stream_t stream);
int ali_bn_bwd(const float* x, const float* g, const float* mask_in, const float* mask_pre,
               const float* mean, const float* invstd, const float* gamma,
               int32_t B, int32_t rows_per_img, int32_t C, int32_t batch_stats, float lrelu_slope,
               float* dgamma, float* dbeta, float* gx, void* ws, size_t ws_bytes, ali_stream_t stream);

/* The same two ops fed by the per-tile partial sums a convolution launch left in `part` (AliEpilogue.bn_part, slots =
 * M-tiles of that launch): no reduction pass over the tensor.  stats: training mode only, `count` = rows per pass
 * (B/groups * H*W).  bwd: dgamma / dbeta from the partials, then the elementwise gx pass of ali_bn_bwd. */
int ali_bn_stats_from_partials(const float* part, int32_t slots, int32_t groups, int32_t C, int64_t count,
                               const float* gamma, const float* beta, float* running_mean, float* running_var,
                               float momentum, float eps, float* mean, float* invstd, float* sc, float* sh,
                               int64_t stat_stride, ali_stream_t stream);
int ali_bn_bwd_from_partials(const float* part, int32_t slots, const float* x, const float* g, const float* mask_in,
                             const float* mask_pre, const float* mean, const float* invstd, const float* gamma,
                             int32_t B, int32_t rows_per_img, int32_t C, int32_t batch_stats, float lrelu_slope,
                             float* dgamma, float* dbeta, float* gx, ali_stream_t stream);

/* nn.BCEWithLogitsLoss (mean) against a constant target, forward + gradient,
 * and the diagnostic sigmoid().mean() (mnist.py:181,228-248).
 * out[0] = loss, out[1] = mean(sigmoid(logit)); glogit[i] = gscale*(sigmoid-t)/B. */
int ali_bce_logits(const float* logit, int32_t B, float target, float gscale, float* out2, float* glogit,
                   ali_stream_t stream);

/* torch.optim.Adam step (mnist.py:176-179,230,236,241), no amsgrad / decay.
 * One launch over a flat parameter segment.  The 1-based step count is `step`, or *dev_step when
 * dev_step != NULL (graph replays); the gradient is read as grad_scale * g (1/world for DP).
 * With `arrive` (a zeroed device int32 the launch leaves at zero; needs dev_step): *dev_step is the number of COMPLETED
 * steps -- the launch runs step *dev_step + 1 and its last block stores that value back, so a captured graph's step
 * count advances without a launch of its own.  p16 (optional, n halves): also leave (_Float16)p -- the fp16 twin of
 * master weights that already live in a GEMM's layout (AliEpilogue.w16), so that no re-pack launch has to produce it. */
int ali_adam(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1, float beta2,
             float eps, int32_t step, int32_t* dev_step, int32_t* arrive, float grad_scale, void* p16,
             ali_stream_t stream);
/* counters[i][0] += incs[i] for n device int64 counters in one launch (nn.BatchNorm2d.num_batches_tracked of every
 * layer, mnist.py:111-123 forward in train mode; the stepper's iteration counter). */
int ali_add_i64_multi(int32_t n, int64_t* const* counters, const int64_t* incs, ali_stream_t stream);

/* BCEWithLogitsLoss of two passes batched along the rows, rows [0,B) against target_a and [B,2B) against target_b
 * (mnist.py:228: (bce(D_valid, 0) + bce(D_fake, 1)) / 2; :245-248: the two sigmoid().mean() scores):
 * out3 = {(loss_a + loss_b)/2, mean sigmoid(a), mean sigmoid(b)}; glogit[i] = gscale*(sigmoid - target)/B or NULL. */
int ali_bce_logits_pair(const float* logit, int32_t B, float target_a, float target_b, float gscale, float* out3,
                        float* glogit, ali_stream_t stream);

/* Attribute plumbing of one batch (mnist.py:47-55, audio_mnist.py:203-210, whalecalls.py:455): idx[b*n_cat + j] =
 * argmax of categorical attribute j (one-hot rows of n_classes[j] floats, or int32 when cat_is_int[j]; first maximum
 * like torch.argmax); cont[b*n_cont + j] = cont_in[j][b]. */
int ali_attr_pack(const void* const* cat, const int32_t* n_classes, const int32_t* cat_is_int, int32_t n_cat,
                  const float* const* cont_in, int32_t n_cont, int32_t B, int32_t* idx, float* cont, ali_stream_t stream);

/* Generator input (mnist.py:76-85; audio_mnist.py:250-256): out[b*ld + :] = [ z[b, :zdim] | onehot_j[b] @ tables[j]
 * ([n_classes[j]][256]) for j < n_emb | cont[b, :n_cont] | zeros ] -- a true sum over classes (soft attributes), the
 * zero padding brings the row to the GEMM's channel stride.  ali_g_input_table_grad: the gradient of one table,
 * out[n][k] = sum_b onehot[b][n] * g[b*ld + off + k] (samples in order, no atomics). */
int ali_g_input(const float* z, int32_t zdim, const void* const* onehot, const int32_t* n_classes,
                const int32_t* onehot_is_int, const float* const* tables, int32_t n_emb, const float* cont, int32_t n_cont,
                int32_t B, int32_t ld, float* out, ali_stream_t stream);
int ali_g_input_table_grad(const void* onehot, int32_t onehot_is_int, int32_t n_classes, const float* g, int32_t ld,
                           int32_t off, int32_t B, float* out, ali_stream_t stream);

/* Conditioning-plane assembly (mnist.py:24-29,46-55; audio_mnist.py:178-209):
 * out[b,h,w,0] = X[b,h,w]; out[..,1+j] = tanh(emb_j[idx_j[b]][src(h,w)]) for the
 * n_emb categorical planes (16x16 tables, nearest up-sampling, src = floor(dst*16/H));
 * then n_cont broadcast scalars cont[b,j]; remaining channels up to Cpad = 0. */
int ali_assemble_planes(const float* X, const int32_t* idx, const float* const* emb_tables, int32_t n_emb,
                        const float* cont, int32_t n_cont, float* out, int32_t B, int32_t H, int32_t W,
                        int32_t Cpad, const float* mask /* optional [B][mask_ld]: out[b,..,c] *= mask[b][c], the
                        Dropout2d in front of the consuming conv (mnist.py:118) */, int32_t mask_ld, ali_stream_t stream);

/* Backward of ali_assemble_planes for one embedding table (mnist.py:17-18,46-55 and copies: Embedding -> 16x16 ->
 * nearest Upsample -> Tanh):  out[n][cell] = sum over samples b with idx[b*idx_ld + idx_col] == n and over the pixels
 * (h,w) that the up-sampling maps to cell (floor(h*16/H)*16 + floor(w*16/W)) of
 *   g[((b*H+h)*W+w)*g_ld + g_ch] * (1 - x[((b*H+h)*W+w)*x_ld + x_ch]^2)        (tanh' from the stored plane).
 * Fixed summation order (no atomics).  out is [n_rows][256]. */
int ali_plane_table_grad(const float* g, int32_t g_ld, int32_t g_ch, const float* x, int32_t x_ld, int32_t x_ch,
                         const int32_t* idx, int32_t idx_ld, int32_t idx_col, int32_t B, int32_t H, int32_t W,
                         int32_t n_rows, float* out,
                         const float* table /* optional [n_rows][256]: take the plane value tanh(table[n][cell]) from
                         the table instead of x (x may then be NULL): the stored planes may carry a Dropout2d mask */,
                         ali_stream_t stream);

/* Gather half of a strided transposed convolution in scatter form (ConvTranspose2d forward with one or two output
 * channels -- audio_mnist.py:281, whalecalls.py / esrf_acoustic.py Generator tails -- and the few input planes of a
 * first Conv2d's data gradient that are consumed, audio_mnist.py:203-210): a 1x1 GEMM (ali_conv_fwd) first produces,
 * for every pixel (b,h,w) of the H x W map, the contribution contrib[(b,h,w)*ldc + (r*S+s)*NC + c] of that pixel to
 * output channel c through tap (r,s); this sums, per output pixel, the taps whose source position exists:
 *   out[((b*Hout+oy)*Wout+ox)*ostride + c] = act(bias[c] + sum_{r,s} contrib[b,(oy+pad-r)/stride,(ox+pad-s)/stride,c,r,s])
 * over the (r,s) for which both quotients are exact and inside the map.  NC <= 8. */
int ali_col2im(const float* contrib, int32_t ldc, const float* bias, float* out, int32_t B, int32_t H, int32_t W,
               int32_t Hout, int32_t Wout, int32_t NC, int32_t ostride, int32_t R, int32_t S, int32_t stride,
               int32_t pad, int32_t act, float slope, ali_stream_t stream);
/* The two halves above in ONE launch, for 64-channel maps and at most two output channels (what the spectrogram stacks
 * have at their one-channel ends): x [B,H,W,64] times the tap matrix w_nc [NC*R*S][64] (row (r*S+s)*NC + c, the layout
 * the 1x1 GEMM takes) on the matrix cores in exact fp32, contributions kept in LDS per output tile, then the sum and
 * epilogue of ali_col2im -- the [pixels][taps] tensor is never written.  ali_tconv_scatter_ok tells whether a shape is
 * served (otherwise: ali_conv_fwd + ali_col2im). */
int32_t ali_tconv_scatter_ok(int32_t C, int32_t NC, int32_t R, int32_t S, int32_t stride);
int ali_tconv_scatter(const float* x, const float* w_nc, const float* bias, float* out, int32_t B, int32_t H, int32_t W,
                      int32_t C, int32_t Hout, int32_t Wout, int32_t NC, int32_t ostride, int32_t R, int32_t S,
                      int32_t stride, int32_t pad, int32_t act, float slope, ali_stream_t stream);

/* Weight gradient of that transposed convolution for ONE output channel: dw[k*s_k + (r*S+s)*s_tap] = sum_{b,p,q}
 * big[b,p,q,k] * small[(b, p*stride - pad + r, q*stride - pad + s) * sstride], big = its 64-channel input [B,P,Q,64], small
 * = the gradient of its [B,H,W] output (element stride sstride).  One launch + a slab fold; ali_tconv_scatter_wgrad_ws
 * returns the workspace bytes it needs, 0 when the shape is not served (then: ali_conv_bwd_weight). */
int64_t ali_tconv_scatter_wgrad_ws(int32_t B, int32_t P, int32_t Q, int32_t K, int32_t R, int32_t S, int32_t stride);
int ali_tconv_scatter_wgrad(const float* big, const float* small, int32_t sstride, float* dw, int64_t s_k, int64_t s_tap,
                            int32_t B, int32_t P, int32_t Q, int32_t K, int32_t H, int32_t W, int32_t R, int32_t S,
                            int32_t stride, int32_t pad, void* ws, size_t ws_bytes, ali_stream_t stream);

/* Tail of the spectrogram front-end (the step in front of the path): torchaudio.transforms.Spectrogram
 * (power 2) + (. + 1e-6).log() and, optionally, spect_to_img (audio_mnist.py:116,347-363 and copies).  `y` holds, per
 * frame (b,t), the windowed DFT as produced by a 1x1 ali_conv_fwd with the [2F x win] cos|sin matrix: re[f] = y[f],
 * im[f] = y[F+f].  out[b,f,t] = log(re^2 + im^2 + 1e-6); with mean/std (per last-dim index t, as the reference
 * computes them): clip((. - mean[t]) / (std[t] + 1e-6), -k, k) / k. */
int ali_spect_post(const float* y, int32_t B, int32_t T, int32_t F, const float* mean, const float* std, float clip_k,
                   float* out, ali_stream_t stream);

const char* ali_last_error(void);
/* The Discriminator's one-output head, Conv2d(C, 1, 1) on a 1x1 map (mnist.py:127), as a GEMV:
 *   ali_head_fwd  : y[b] = bias[0] + sum_c x[b][c] * w[c]            (x rows ld floats apart, C % 4 == 0)
 *   ali_head_wgrad: dw[c] = sum_b g[b] * x[b][c],  db[0] = sum_b g[b] (db optional)                          */
int ali_head_fwd(const float* x, int32_t ld, const float* w, const float* bias, float* y, int32_t B, int32_t C,
                 ali_stream_t stream);
int ali_head_wgrad(const float* x, int32_t ld, const float* g, float* dw, float* db, int32_t B, int32_t C,
                   ali_stream_t stream);
/* n device-to-device copies (dst[i] <- src[i], bytes[i] each; pointers and sizes multiples of 4) in one launch per 8:
 * the per-step refresh of the input buffers a captured HIP graph reads. */
int ali_copy_multi(int32_t n, const void* const* src, void* const* dst, const int64_t* bytes, ali_stream_t stream);
int ali_version(void);
/* Re-read the developer tuning variables (ALI_SPLITK, ALI_NO_ORDER, ...: csrc/ali_common.h) from the environment;
 * they are otherwise read once per process.  Tests and sweeps only. */
void ali_reload_tuning(void);

#ifdef __cplusplus
}
#endif
#endif /* ALI_HIP_H */

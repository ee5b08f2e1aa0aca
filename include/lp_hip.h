/* lp_hip.h -- C ABI of the MI355X (gfx950) latent-pose hot-path kernels.
 *
 * Drop-in boundary: the reference has no native layer -- its hot path is ATen calls issued from
 * Python nn.Modules (generators/common/blocks.py, generators/vector_pose_unsupervised_segmentation_noBottleneck.py).
 * Each entry point below replaces the group of ATen calls named in its comment.  Conventions:
 *   - plain C: raw DEVICE pointers, explicit int sizes, scalar hyper-parameters, a hipStream_t passed as void*;
 *   - the caller (PyTorch) owns every buffer, including scratch; nothing is allocated or freed here;
 *   - all work is enqueued asynchronously on `stream`; no hidden synchronisation, no global mutable state;
 *   - return value 0 = ok, negative = LP_ERR_*; lp_last_error() gives a thread-local message;
 *   - activations are NHWC fp32 (torch channels_last storage); conv operands are 16-bit planes (lp_act_pack, lp_pack_weights);
 *     `prec` selects the operand format: LP_PREC_BF16 | LP_PREC_F16 (1 MFMA / k-step) or LP_PREC_BF16X3 (hi+lo split, fp32-class).
 */
#ifndef LP_HIP_H
#define LP_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define LP_OK 0
#define LP_ERR_ARG (-1)
#define LP_ERR_UNSUPPORTED (-2)
#define LP_ERR_HIP (-3)

#define LP_PREC_BF16 0
#define LP_PREC_BF16X3 1
#define LP_PREC_F16 2

const char* lp_last_error(void);
int lp_abi_version(void);

/* Re-layout + 16-bit conversion of a conv/linear weight.  w: fp32 [Cout][Cin][T] (reference nn.Conv2d layout, T = k*k).
 * mode 0 (forward):  out[t][co][ci] = w[co][ci][t]            rows padded to CoutP (x128), cols to CinP (x32)
 * mode 1 (dgrad):    out[T-1-t][ci][co] = w[co][ci][t]        rows = Cin padded to RowsP, cols = Cout padded to ColsP
 * modes 2 / 3 (ABI 10; T = 9 in, SIXTEEN taps out, t = phase * 4 + i * 2 + j): the phase form of nn.Upsample(scale_factor=2) + 3x3 conv
 *   (generators/common/blocks.py:74-88).  Output pixel (2y + a, 2x + b), phase = 2a + b, only meets the low-resolution pixels
 *   (y + a - 1 + i, x + b - 1 + j); the taps that meet the same pixel are summed in fp32 before the 16-bit split.
 *   mode 2 (forward): out[t][co][ci];  mode 3 (data gradient): out[t][ci][co] = the (phase, 1 - i, 1 - j) sums.
 *   Consumed by lp_conv16_fwd(_stats) with upsample = 2 (forward) / 3 (data gradient): 4/9 of the matrix work of the fused-upsample conv.
 * f16 = 0: hi = bf16(w), lo = bf16(w - hi) (may be NULL);  f16 = 1: hi = fp16(w) (saturating), lo unused.
 * Replaces nothing in the reference (cuDNN does this internally). */
int lp_pack_weights(const float* w, uint16_t* hi, uint16_t* lo, int Cout, int Cin, int T, int RowsP, int ColsP, int mode, int f16,
                    void* stream);
/* batched: table = DEVICE array of {const float* w; uint16_t* hi; uint16_t* lo; int Cout, Cin, T, RowsP, ColsP, mode, chunk0, f16;}
 * (lp_pack_desc_bytes() each); one launch packs every (weight, orientation) entry -- all convs of a module after an optimizer step.
 * The grid is flat over 1024-element chunks: chunk0 = sum of ceil(T'*RowsP*ColsP/1024) of the preceding entries (ascending; T' = 16 for modes 2 / 3),
 * total_chunks = that sum over all entries. */
int lp_pack_desc_bytes(void);
int lp_pack_weights_batch(const void* table, int num_entries, long long total_chunks, void* stream);
/* Forward AND data-gradient pack of a conv weight (T = 1 | 9) from one coalesced read.  table: DEVICE array of {const float* w;
 * uint16_t* hi0, *lo0 (forward pack, rows RowsP0 x cols ColsP0), *hi1, *lo1 (data-gradient pack, RowsP1 x ColsP1); int Cout, Cin, T,
 * RowsP0, ColsP0, RowsP1, ColsP1, tile0, tiles_ci, f16;} (lp_pack_pair_desc_bytes() each); an entry owns the 32 x 32 tiles
 * [tile0, tile0 + tiles_co * tiles_ci) with tiles_co = ceil(max(RowsP0, ColsP1) / 32), tiles_ci = ceil(max(ColsP0, RowsP1) / 32). */
int lp_pack_pair_desc_bytes(void);
int lp_pack_weights_pairs(const void* table, int num_entries, long long total_tiles, void* stream);

/* Operand planes of a conv input: hi (, lo) [N*HW][C8] 16-bit, C8 = C rounded up to 8 (pad channels zero), holding
 *   act(x) * in_scale,  act: pro 0 identity | 1 relu(x*scale[n,c]+shift[n,c]) | 2 relu(x) | 3 relu6(x*scale[c]+shift[c]) |
 *                            4 relu(x*scale[c]+shift[c]) | 5 x*scale[c]+shift[c]   (3..5: BatchNorm folded to a per-channel affine)
 * in the operand format of `prec` (bf16 | bf16 hi+lo | fp16, saturating).  Replaces, once per tensor, the instance_norm + mul +
 * add + relu chain of AdaptiveNorm2d/ReLU (generators/common/blocks.py:18-26,70-73) that the reference runs before every conv;
 * the planes feed lp_conv16_fwd (forward / dgrad) and lp_conv16_wgrad.  in_scale: device scalar|NULL.
 * fp16 gradient operands: amax_part = amax_count partial maxima of |x|, amax_stride floats apart (|NULL) -- the lp_amax_blocks()
 * contiguous block maxima written by lp_amax_partial, or the lp_amax_slots() slots, lp_amax_slot_stride() floats apart, the kernel
 * that produced x folded them into (the `amax_slots` argument of lp_conv16_fwd, lp_adain_relu_bwd, lp_sum2x2, lp_head_bwd,
 * lp_avgpool2_bwd, lp_l1_bwd: a zeroed buffer of slots * stride floats); the pack then
 * scales by s = the power of two that puts amax(x) into [2^12, 2^13) and writes scale_out = {s, 1/s} (device, |NULL) for the
 * consumers (alpha2 of lp_conv16_fwd, out_scale of lp_conv16_wgrad). */
int lp_act_pack(const float* x, const float* scale, const float* shift, int pro, uint16_t* hi, uint16_t* lo,
                int N, int HW, int C, int prec, const float* in_scale, const float* amax_part, int amax_count, int amax_stride,
                float* scale_out, void* stream);
int lp_amax_blocks(void);
int lp_amax_slots(void);
int lp_amax_slot_stride(void);
int lp_amax_partial(const float* x, long long numel, float* part, void* stream);

/* Fused conv on operand planes: y = alpha * alpha2 * conv_{k x k, pad k/2}( up2?(a), w ) + bias + res
 * Replaces, per conv of blocks.ResBlock (generators/common/blocks.py:70-111): nn.Upsample(nearest x2) + F.conv2d + W/sigma scaling
 * (spectral_norm) + residual add; with the mode-1 pack and a = dY it is the data-gradient kernel.
 *   a_hi/a_lo [N][H/(up?2:1)][W/(up?2:1)][C8] (lp_act_pack), y [N][H][W][Cout] fp32, bias [Cout]|NULL,
 *   res [N][H>>res_shift][W>>res_shift][Cout]|NULL, alpha, alpha2: device scalars|NULL (=1; 1/sigma and 1/in_scale).  ksize 1|3.
 *   relu_mask16 [N][H][W][Co8]|NULL: 16-bit activation plane; y is zeroed where it is <= 0 -- the backward of the ReLU that produced
 *   the operand of the conv whose data gradient this launch computes (replaces a dx = dA * (x > 0) pass; blocks.py:71-73,84).
 *   out_hi/out_lo [N][H][W][Co8]|NULL: also emit the operand planes of (out_relu ? relu(y) : y) for the consumer conv.
 *   workspace (lp_conv16_fwd_workspace_bytes(); 0 = never needed) | NULL: split-K partial tiles for the layers whose output tiling
 *   cannot fill the chip (4x4 .. 16x16 maps); without it those layers run unsplit.
 *   upsample (ABI 10): 0 | 1 (fused nearest x2 in front of the 3x3 conv) | 2: the same conv in its PHASE form -- w = an lp_pack_weights mode-2
 *   image (16 taps), per output phase a 2 x 2 conv on the low-resolution planes: identical result up to the fp32 summation of coinciding taps,
 *   4/9 of the matrix work | 3: the phase form of its DATA GRADIENT -- a = dy planes [N][2H][2W][C8], w = a mode-3 image, y = the gradient
 *   w.r.t. the LOW-resolution input [N][H][W][Cout] (the 2 x 2 sum of the upsample's adjoint included): replaces the 3x3 data-gradient conv on
 *   the 2H x 2W grid + lp_sum2x2 / the upsample flag of lp_adain_relu_bwd. */
int lp_conv16_fwd(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* w_hi, const uint16_t* w_lo, float* y,
                  const float* bias, const float* res, const float* alpha, const float* alpha2,
                  int N, int H, int W, int Cin, int Cout, int CinP, int CoutP,
                  int ksize, int upsample, int res_shift, int prec, const uint16_t* relu_mask16,
                  uint16_t* out_hi, uint16_t* out_lo, int out_relu, float* workspace, long long workspace_bytes, float* amax_slots,
                  void* stream);
long long lp_conv16_fwd_workspace_bytes(int N, int H, int W, int Cout, int ksize);
/* lp_conv16_fwd with two more outputs options (blocks.py:18-26: the instance norm that follows every generator conv; the BatchNorm of the
 * embedder convs):  y may be NULL when out_hi is given and Cout % 8 == 0 (16-bit activation residency: no fp32 store);
 * stats [rows][Cout][3] (capacity stats_capacity_floats >= lp_conv16_stats_floats()) | NULL: {count, mean, M2} of the written values per
 * (64-pixel row block, channel) from the conv epilogue -- no extra pass over y; *stats_rows (HOST int) = row blocks per image, or 0 when
 * the geometry is not covered (maps under 64 pixels, ragged tiles, split-K): then run lp_instnorm_stats / lp_bn_train_stats on y.
 * lp_norm_stats_finalize merges them: mean, rstd, scale = gamma*rstd, shift = beta - mean*scale per (n, c) (+ BatchNorm running stats, N = 1). */
int lp_conv16_fwd_stats(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* w_hi, const uint16_t* w_lo, float* y,
                        const float* bias, const float* res, const float* alpha, const float* alpha2,
                        int N, int H, int W, int Cin, int Cout, int CinP, int CoutP,
                        int ksize, int upsample, int res_shift, int prec, const uint16_t* relu_mask16,
                        uint16_t* out_hi, uint16_t* out_lo, int out_relu, float* workspace, long long workspace_bytes, float* amax_slots,
                        float* stats, long long stats_capacity_floats, int* stats_rows, void* stream);
long long lp_conv16_stats_floats(int N, int H, int W, int Cout);
int lp_norm_stats_finalize(const float* part, int S, const float* gamma, const float* beta, int ab_stride, float eps, float momentum,
                           float* running_mean, float* running_var, float* mean, float* rstd, float* scale, float* shift,
                           int N, int C, void* stream);

/* Weight gradient: dw[co][ci][t] = out_scale * sum_{n,y,x} dy[n,y,x,co] * up2?(a)[n,y+dy_t,x+dx_t,ci]  (autograd of F.conv2d
 * w.r.t. weight, blocks.py:76-88) on operand planes: a = the planes the forward conv consumed, dy = lp_act_pack of the output
 * gradient.  Two launches: partial slabs over `splits` pixel ranges, then a reduction that also writes the reference
 * [Cout][Cin][k][k] layout.  workspace: lp_conv_wgrad_workspace_bytes().
 * dbias [Cout]|NULL: also emit out_scale * sum_{n,y,x} dy[n,y,x,co] (the kernel streams dy anyway); dbias_accumulate: ADD it to dbias
 * (the parameter's .grad) instead of storing.  out_scale: device scalar|NULL.
 * sn_w_orig [Cout][Cin][k][k] + sn_dot [lp_conv_wgrad_dot_blocks()] (both or neither): spectrally normalised layer -- the reduction
 * also leaves per-block partial sums of <dw, W_orig> in sn_dot, which lp_sn_grad_apply(ndot = that count) consumes. */
long long lp_conv_wgrad_workspace_bytes(int Cin, int Cout, int ksize, int splits);
int lp_conv16_wgrad(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* dy_hi, const uint16_t* dy_lo, float* dw,
                    float* workspace, int N, int H, int W, int Cin, int Cout, int ksize, int upsample, int splits, int prec,
                    float* dbias, int dbias_accumulate, const float* out_scale, const float* sn_w_orig, float* sn_dot, void* stream);
int lp_conv_wgrad_dot_blocks(int Cin, int Cout, int ksize);

/* Thin-channel convs (<= 4 channels on one side: RGB -> 64 first convs of the critics / VGG stacks, the generator head's weight
 * gradient): bandwidth-bound kernels on plain NHWC fp32 activations (no input operand planes); weights from the same packs.
 *   lp_thin_conv_fwd:  y = alpha * conv(x, w) + bias for Cin <= 4, Cout % 64 == 0 (lp_thin_conv_supported).  fp32 VALU kernel; the
 *                      RGB 3x3 case in the fp16 / bf16 modes (lp_thin_conv_emits_planes: Cin 3, W % 16 == 0) runs as one MFMA k-step
 *                      per output block and can also write out_hi [N][H][W][Cout] = the operand planes of (out_relu ? relu(y) : y);
 *                      y may then be NULL (planes-only output, ABI 7).
 *   lp_thin_wgrad:     dw (and dbias when lp_thin_wgrad_has_dbias) for Cin <= 4 (pro 0) or Cout <= 4 (3x3; the AdaIN/ReLU prologue
 *                      pro/scale/shift of the wide input is applied on the fly); workspace as lp_conv_wgrad_workspace_bytes(). */
int lp_thin_conv_supported(int Cin, int Cout, int ksize, int W);
int lp_thin_conv_fwd(const float* x, const uint16_t* w_hi, const uint16_t* w_lo, float* y, const float* bias, const float* alpha,
                     int N, int H, int W, int Cin, int Cout, int CinP, int CoutP, int ksize, int prec, uint16_t* out_hi, int out_relu,
                     void* stream);
int lp_thin_conv_emits_planes(int Cin, int Cout, int ksize, int W, int prec);
int lp_thin_wgrad_supported(int Cin, int Cout, int ksize, int pro, int W);
int lp_thin_wgrad_has_dbias(int Cin, int Cout);
int lp_thin_wgrad(const float* x, const float* dy, float* dw, float* workspace, const float* scale, const float* shift,
                  int N, int H, int W, int Cin, int Cout, int ksize, int pro, int splits, float* dbias, void* stream);

/* Small-batch linear layer y = alpha * x W^T + bias, x [B][K], W [N][K] (nn.Linear layout), 1 <= B <= 64, K % 4 == 0, K <= 1024:
 * the AdaIN-parameter projector (noBottleneck.py:96-101) and the critic's 512 -> 1 head (no_landmarks.py:88,105) -- weight streams,
 * not GEMMs.  alpha: device scalar|NULL (1/sigma of the spectral norm), bias [N]|NULL.
 * Backward: dx [B][K] = alpha * g W (|NULL), dw [N][K] = g^T x (RAW gradient w.r.t. W/sigma: lp_sn_grad_apply turns it into the
 * gradient w.r.t. W_orig; |NULL), db [N] = column sums of g (|NULL); workspace: lp_linear_bwd_workspace_bytes(). */
int lp_linear_fwd(const float* x, const float* w, const float* bias, const float* alpha, float* y, int B, int N, int K, void* stream);
long long lp_linear_bwd_workspace_bytes(int B, int N, int K);
int lp_linear_bwd(const float* x, const float* w, const float* g, const float* alpha, float* dx, float* dw, float* db,
                  float* workspace, int B, int N, int K, void* stream);

/* criterions/idt_embed.py:58-83 crop_and_resize: images [N][C][H][W] (NCHW fp32), boxes [N][4] = t, b, l, r in pixels ->
 * out [N][C][Ho][Wo]: affine_grid(align_corners=False) + grid_sample(bilinear, reflection padding).  The backward pass is the adjoint
 * (dimages is zeroed, then accumulated with fp32 atomics). */
int lp_grid_crop_fwd(const float* images, const float* boxes, float* out, int N, int C, int H, int W, int Ho, int Wo, void* stream);
int lp_grid_crop_bwd(const float* dout, const float* boxes, float* dimages, int N, int C, int H, int W, int Ho, int Wo, void* stream);

/* MobileNetV2 pose encoder, forward (embedders/unsupervised_pose_separate_embResNeXt_segmentation.py:26-28,56-58 = torchvision
 * mobilenet_v2(num_classes)), fp32: stem, depthwise and pointwise convs with the producer's BatchNorm (+ ReLU6, + residual) applied
 * while the input is loaded and the BatchNorm statistics of the output folded into the same launch.  (lp_affine_res / lp_act_pack
 * pro 3 + lp_conv16_fwd is the alternative MFMA route for the 1x1 convs.)  BatchNorm enters as per-channel
 * (scale, shift): running statistics in eval mode; lp_bn_stats in train mode.
 *   lp_stem_conv_s2:  x [N][3][H][W] NCHW fp32, w [Cout][3][3][3] -> y [N][H/2][W/2][Cout]   (3x3, stride 2, pad 1, no bias)
 *   lp_dwconv3x3_fwd: depthwise 3x3 pad 1 stride 1|2 on relu6(x*in_scale[c]+in_shift[c]) (in_scale NULL: on x), w [C][3][3]
 *   lp_affine_res:    x = y*scale[c]+shift[c] (+res) over P positions; hi/lo|NULL: also the operand planes [P][C] of x (C % 8 == 0)
 *   lp_affine_relu6_mean: out [N][C] = mean over HW of relu6(y*scale[c]+shift[c])            (features[18] BN + ReLU6 + avg-pool)
 *   lp_bn_stats:      train-mode nn.BatchNorm2d over y [P][C] (P = N*H*W): scale = gamma/sqrt(var+eps), shift = beta - mean*scale
 *                     (biased batch variance), running_mean/var|NULL updated with `momentum` (unbiased variance);
 *                     workspace: lp_bn_stats_workspace_bytes(P, C) */
int lp_stem_conv_s2(const float* x, const float* w, float* y, int N, int H, int W, int Cout, void* stream);
int lp_dwconv3x3_fwd(const float* x, const float* w, const float* in_scale, const float* in_shift, float* y,
                     int N, int H, int W, int C, int stride, void* stream);
int lp_affine_res(const float* y, const float* scale, const float* shift, const float* res, float* x, uint16_t* hi, uint16_t* lo,
                  long long P, int C, int prec, void* stream);
int lp_affine_relu6_mean(const float* y, const float* scale, const float* shift, float* out, int N, int HW, int C, void* stream);
long long lp_bn_stats_workspace_bytes(long long P, int C);
/*   lp_pwconv_fwd:    1x1 conv on the VALU in fp32 (the encoder's layers are launch- and bandwidth-bound, not MFMA work):
 *                     y [P][N] = a w^T, a = (in_relu6 ? relu6 : id)(x*in_scale[k]+in_shift[k]) (+ in_res [P][K]); w [N][K] (nn.Conv2d
 *                     layout); in_scale/in_shift|NULL (both or neither); x_out|NULL: `a` written back (the block input a later
 *                     residual needs); stats_part|NULL: 9 * lp_pwconv_stat_rows(P, K, N) * N/4 floats, BatchNorm partials of y
 *   lp_dwconv3x3_stats_fwd: lp_dwconv3x3_fwd + the BatchNorm partials of its output (9 * lp_dwconv_stat_rows() * C/4 floats)
 *   lp_bn_finalize:   (scale, shift) + running-statistics update from `rows` partials per channel group */
int lp_pwconv_stat_rows(long long P, int K, int N);
int lp_pwconv_fwd(const float* x, const float* w, float* y, const float* in_scale, const float* in_shift, int in_relu6,
                  const float* in_res, float* x_out, float* stats_part, long long P, int K, int N, void* stream);
int lp_dwconv_stat_rows(int N, int H, int W, int C, int stride);
int lp_dwconv3x3_stats_fwd(const float* x, const float* w, const float* in_scale, const float* in_shift, float* y, float* stats_part,
                           int N, int H, int W, int C, int stride, void* stream);
int lp_bn_finalize(const float* stats_part, int rows, const float* gamma, const float* beta, float* running_mean, float* running_var,
                   float* scale, float* shift, int C, float eps, float momentum, void* stream);
int lp_bn_stats(const float* y, const float* gamma, const float* beta, float* running_mean, float* running_var, float* scale,
                float* shift, float* workspace, long long P, int C, float eps, float momentum, void* stream);

/* MobileNetV2 backward (meta-training trains the pose encoder, runners/holycow.py:34-41): the depthwise 3x3 conv's data gradient
 * w.r.t. its activated input (da [N][H][W][C] from dy [N][Ho][Wo][C]) and weight gradient dw [C][3][3] (the activation
 * relu6(x*in_scale[c]+in_shift[c]) of the raw input x is recomputed on load, in_scale NULL: identity);
 * workspace: lp_dwconv3x3_wgrad_workspace_bytes(C).  The 1x1 convs, BatchNorm and ReLU6 backward run on lp_conv16_fwd /
 * lp_conv16_wgrad / lp_norm_act_bwd (act_hi = 6). */
int lp_dwconv3x3_dgrad(const float* dy, const float* w, float* da, int N, int H, int W, int C, int stride, void* stream);
long long lp_dwconv3x3_wgrad_workspace_bytes(int C);
int lp_dwconv3x3_wgrad(const float* x, const float* in_scale, const float* in_shift, const float* dy, float* dw, float* workspace,
                       int N, int H, int W, int C, int stride, void* stream);

/* Instance-norm statistics of x [N][H*W][C] and the AdaIN scale/shift derived from them (blocks.py:18-26):
 *   mean/rstd [N][C] (biased variance, eps), scale = rstd*gamma, shift = beta - mean*scale.
 * gamma/beta [N][C] with row stride `ab_stride` floats (they are slices of the projector output, noBottleneck.py:108-125).
 * workspace: lp_instnorm_workspace_bytes(). */
long long lp_instnorm_workspace_bytes(int N, int HW, int C);
int lp_instnorm_stats(const float* x, const float* gamma, const float* beta, int ab_stride, float eps,
                      float* mean, float* rstd, float* scale, float* shift, float* workspace,
                      int N, int HW, int C, void* stream);

/* Backward of relu(AdaIN(x)) [+ nearest x2 upsample]: (autograd of blocks.py:18-26,73-75)
 *   g = sum2x2?(dA) * [x*scale+shift > 0];  dgamma = sum g*xhat;  dbeta = sum g;
 *   dx = gamma*rstd*(g - mean(g) - xhat*mean(g*xhat)) (+ add)
 * dA [N][H<<up][W<<up][C], x/dx/add [N][H][W][C], dgamma/dbeta [N][C] with row stride ab_stride (written, not accumulated).
 * workspace: lp_adain_bwd_workspace_bytes(). */
long long lp_adain_bwd_workspace_bytes(int N, int HW, int C);
int lp_adain_relu_bwd(const float* dA, const float* x, const float* add, const float* gamma, int ab_stride,
                      const float* mean, const float* rstd, const float* scale, const float* shift,
                      float* dx, float* dgamma, float* dbeta, float* workspace,
                      int N, int H, int W, int C, int upsample, float* amax_slots, void* stream);

/* Generalisation used by the embedder's BatchNorm layers (a train-mode BatchNorm over [P][C] is this with N = 1, H*W = P, ab_stride = C):
 *   mask_mode 0: g = dA * [0 < x*scale+shift < act_hi]  (ReLU: act_hi <= 0 or huge; ReLU6: 6);  1: g = dA (no activation);
 *             2: g = dA * [mask_src > 0]  (the ReLU sits behind a residual add; mask_src [N][H][W][C] = the block output)
 *   g_copy [N][H][W][C]|NULL: also receives g (the identity branch's gradient);  frozen_stats = 1: mean/rstd are constants
 *   (eval-mode BatchNorm on running statistics): dx = gamma*rstd*g. */
int lp_norm_act_bwd(const float* dA, const float* x, const float* add, const float* gamma, int ab_stride,
                    const float* mean, const float* rstd, const float* scale, const float* shift,
                    float* dx, float* dgamma, float* dbeta, float* workspace,
                    int N, int H, int W, int C, int upsample, int mask_mode, const float* mask_src, float* g_copy, float act_hi,
                    int frozen_stats, float* amax_slots, void* stream);

/* Train-mode nn.BatchNorm2d statistics of y [P][C] (torchvision resnext50_32x4d / mobilenet_v2 layers of
 * embedders/unsupervised_pose_separate_embResNeXt_segmentation.py:26-28): mean, rstd = 1/sqrt(biased var + eps), scale = gamma*rstd,
 * shift = beta - mean*scale, all [C]; running_mean/running_var|NULL updated with `momentum` (unbiased variance).
 * workspace: lp_bn_train_stats_workspace_bytes(P, C). */
long long lp_bn_train_stats_workspace_bytes(long long P, int C);
int lp_bn_train_stats(const float* y, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                      float* running_var, float* mean, float* rstd, float* scale, float* shift, float* workspace,
                      long long P, int C, void* stream);

/* ---- ResNeXt-50 32x4d identity encoder (embedders/unsupervised_pose_separate_embResNeXt_segmentation.py:26,37-54) ----
 * The contractions reuse lp_conv16_fwd / lp_conv16_wgrad (1x1 convs on flattened pixels, the 7x7/2 stem on its im2col rows, the
 * classifier); the entries below are the grouped 3x3 conv and the bandwidth-bound layers around the contractions.
 *   lp_gconv16_fwd:   3x3, pad 1, stride 1 conv with C/group_size groups as a block-diagonal conv over aligned 64-channel blocks
 *                     (group_size | 64); a [N][H][W][C] operand planes, weights from lp_pack_grouped ([9][CP][64]), y fp32; with the
 *                     mode-1 pack and a = dY: the data gradient.  alpha2 as lp_conv16_fwd (1/scale of an fp16 gradient operand).
 *   lp_gconv16_wgrad: dw [C][group_size][3][3]; workspace lp_gconv_wgrad_workspace_bytes(C, splits)
 *   lp_pack_grouped:  w [C][group_size][3][3] -> hi(, lo) [9][CP][64]; mode 0 forward, 1 data gradient (flipped, transposed)
 *   lp_im2col_planes: x [N][C][H][W] fp32 NCHW -> operand rows [N*Ho*Wo][K8], K = C*k*k in nn.Conv2d weight order (zero padding)
 *   lp_bn_relu_maxpool_fwd: out [N][Ho][Wo][C] = MaxPool2d(3,2,1)(relu(y*scale[c]+shift[c])), hi/lo|NULL its operand planes,
 *                     idx|NULL [N][Ho][Wo][C] bytes = window position of the maximum;  lp_maxpool_bwd: dA [N][H][W][C] from d_out + idx
 *   lp_bn_add_act:    out = (relu?)(y*scale[c]+shift[c] + (res | res*res_scale[c]+res_shift[c] | 0)) over [P][C], + operand planes
 *                     (ABI 12: out may be NULL when hi is given -- planes only)
 *   lp_subsample2 / lp_zero_stuff2: out[n,i,j] = in[n,2i,2j] / its adjoint on [N][H][W][row_bytes] tensors (H, W = FULL-resolution dims;
 *                     fp32 NHWC or operand planes: row_bytes % 16 == 0);  lp_add_strided2: d[n,2i,2j,:] += s[n,i,j,:] (fp32)
 *   lp_spatial_mean_fwd/bwd: AdaptiveAvgPool2d(1) */
int lp_gconv16_fwd(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* w_hi, const uint16_t* w_lo, float* y, const float* alpha2,
                   int N, int H, int W, int C, int CP, int prec, float* amax_slots, void* stream);
int lp_gconv16_fwd_stats(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* w_hi, const uint16_t* w_lo, float* y, const float* alpha2,
                         int N, int H, int W, int C, int CP, int prec, float* amax_slots, float* stats, long long stats_capacity_floats,
                         int* stats_rows, void* stream);
/*   lp_gconv16_fwd_planes: as lp_gconv16_fwd_stats with y (fp32) OPTIONAL and the operand planes of y (o_hi [, o_lo]) as a second, optional
 *                     output: the fp16 mode keeps the embedder's conv outputs 16-bit resident ("y16": the unscaled fp16 plane, no fp32 y) */
/*                     group_size (ABI 7; 0 = not given): groups of <= 32 channels let the kernel skip the zero half of the block-diagonal product */
int lp_gconv16_fwd_planes(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* w_hi, const uint16_t* w_lo, float* y, uint16_t* o_hi,
                          uint16_t* o_lo, const float* alpha2, int N, int H, int W, int C, int CP, int group_size, int prec, float* amax_slots,
                          float* stats, long long stats_capacity_floats, int* stats_rows, void* stream);
long long lp_gconv_wgrad_workspace_bytes(int C, int splits);
int lp_gconv16_wgrad(const uint16_t* a_hi, const uint16_t* a_lo, const uint16_t* dy_hi, const uint16_t* dy_lo, float* dw, float* workspace,
                     int N, int H, int W, int C, int group_size, int splits, int prec, const float* out_scale, void* stream);
int lp_pack_grouped(const float* w, uint16_t* hi, uint16_t* lo, int C, int group_size, int CP, int mode, int f16, void* stream);
int lp_im2col_planes(const float* x, uint16_t* hi, uint16_t* lo, int N, int C, int H, int W, int ksize, int stride, int pad, int prec,
                     void* stream);
int lp_bn_relu_maxpool_fwd(const float* y, const float* scale, const float* shift, float* out, uint16_t* hi, uint16_t* lo,
                           unsigned char* idx, int N, int H, int W, int C, int prec, void* stream);
int lp_maxpool_bwd(const float* dout, const unsigned char* idx, float* dA, int N, int H, int W, int C, void* stream);
int lp_bn_add_act(const float* y, const float* scale, const float* shift, const float* res, const float* res_scale, const float* res_shift,
                  float* out, uint16_t* hi, uint16_t* lo, long long P, int C, int relu, int prec, void* stream);
/*   lp_bn_add_act_planes (ABI 12, round 6; bf16 / bf16x3): the identity-shortcut form of lp_bn_add_act (torchvision Bottleneck.forward's
 *                     ``out += identity; out = relu(out)``, reached through embedders/unsupervised_pose_separate_embResNeXt_segmentation.py:26,37-54)
 *                     with the residual read from the OPERAND PLANES of the block input (res_hi [+ res_lo]: hi + lo = 16 significant bits, the
 *                     values the block's first conv multiplied) and fp32 ``out`` OPTIONAL (NULL: planes only -- nothing downstream of an
 *                     identity bottleneck of a bf16x3 network reads an fp32 copy of its output) */
int lp_bn_add_act_planes(const float* y, const float* scale, const float* shift, const uint16_t* res_hi, const uint16_t* res_lo, float* out,
                         uint16_t* hi, uint16_t* lo, long long P, int C, int relu, int prec, void* stream);
/*   16-bit-resident conv outputs (fp16 mode; y16 = the fp16 plane [P][C] a conv epilogue wrote instead of fp32 y, C % 8 == 0):
 *   lp_bn_add_act16:  lp_bn_add_act reading y16 (out fp32 + its fp16 operand plane hi|NULL)
 *   lp_bn_act16:      out_hi = fp16((relu?)(y16*scale[c]+shift[c])): the BatchNorm (+ ReLU) prologue of the next conv (lp_act_pack pro 4 / 5
 *                     reading 2 B per element) */
int lp_bn_add_act16(const uint16_t* y16, const float* scale, const float* shift, const float* res, const float* res_scale,
                    const float* res_shift, float* out, uint16_t* hi, long long P, int C, int relu, void* stream);
int lp_bn_act16(const uint16_t* y16, const float* scale, const float* shift, uint16_t* out_hi, long long P, int C, int relu, void* stream);
/*   The generator's 16-bit-resident conv outputs (round 5; fp16 mode: no fp32 copy of a conv output between the AdaIN blocks):
 *   lp_adain_act16:      out_hi = fp16((relu?)(y16[n]*scale[n][c]+shift[n][c])), per-IMAGE affines [N][C] -- AdaptiveNorm2d + ReLU
 *                        (generators/common/blocks.py:18-26,70-73) as the prologue of the next conv, 2 B read + 2 B written per element
 *   lp_adain_relu_bwd16: lp_adain_relu_bwd with x given as that fp16 plane (x-hat and the ReLU pattern from the values the forward normalised)
 *   lp_thin_wgrad16:     lp_thin_wgrad of a conv with <= 4 output channels (the head, noBottleneck.py:80-88) on the fp16 plane of its input */
int lp_adain_act16(const uint16_t* y16, const float* scale, const float* shift, uint16_t* out_hi, int N, long long HW, int C, int relu,
                   void* stream);
int lp_adain_relu_bwd16(const float* dA, const uint16_t* x16, const float* add, const float* gamma, int ab_stride,
                        const float* mean, const float* rstd, const float* scale, const float* shift,
                        float* dx, float* dgamma, float* dbeta, float* workspace,
                        int N, int H, int W, int C, int upsample, float* amax_slots, void* stream);
int lp_thin_wgrad16(const uint16_t* x16, const float* dy, float* dw, float* workspace, const float* scale, const float* shift,
                    int N, int H, int W, int Cin, int Cout, int ksize, int pro, int splits, void* stream);
int lp_subsample2(const void* in, void* out, int N, int H, int W, int row_bytes, void* stream);
int lp_zero_stuff2(const void* in, void* out, int N, int H, int W, int row_bytes, void* stream);
int lp_add_strided2(float* d, const float* s, int N, int H, int W, int C, void* stream);
/*   lp_bn_bwd16:      backward of act(BatchNorm(x)) over x [P][C] written STRAIGHT to the operand planes of dy (hi [, lo]) that the weight /
 *                     data gradient contractions consume -- no fp32 dy, no masked-gradient temporary: out_scale[2] = {s, 1/s} (fp16 mode: the power
 *                     of two taken from a per-channel bound of |dy|; 1 otherwise) goes to alpha2 / out_scale of the consumers; dgamma, dbeta [C];
 *                     mask modes / act_hi / frozen_stats as lp_norm_act_bwd; g_out [P][C]|NULL: the masked incoming gradient in fp32 (the
 *                     identity branch of a residual block); workspace lp_bn_bwd16_workspace_bytes(P, C) */
long long lp_bn_bwd16_workspace_bytes(long long P, int C);
int lp_bn_bwd16(const float* dA, const float* x, const float* mask_src, const float* gamma, const float* mean, const float* rstd,
                const float* scale, const float* shift, uint16_t* out_hi, uint16_t* out_lo, float* out_scale, float* dgamma, float* dbeta,
                float* workspace, long long P, int C, int mask_mode, float act_hi, int frozen_stats, int prec, float* g_out, void* stream);
/*   lp_bn_bwd16_h:    lp_bn_bwd16 with x given as fp32 (x16 = NULL) or as y16 (x = NULL; fp16 mode) and one more mask mode:
 *                     3: mask_src = a 16-bit operand plane [P][C] whose elements are > 0 (the block output's planes instead of its fp32 copy) */
int lp_bn_bwd16_h(const float* dA, const float* x, const uint16_t* x16, const void* mask_src, const float* gamma, const float* mean,
                  const float* rstd, const float* scale, const float* shift, uint16_t* out_hi, uint16_t* out_lo, float* out_scale,
                  float* dgamma, float* dbeta, float* workspace, long long P, int C, int mask_mode, float act_hi, int frozen_stats, int prec,
                  float* g_out, void* stream);
int lp_spatial_mean_fwd(const float* x, float* out, int N, int HW, int C, void* stream);
int lp_spatial_mean_bwd(const float* g, float* dx, int N, int HW, int C, void* stream);

/* out[n,y,x,c] = sum of the 2x2 block of in[n,2y..2y+1,2x..2x+1,c]  (adjoint of nearest x2 upsampling, blocks.py:95). */
int lp_sum2x2(const float* in, float* out, int N, int H, int W, int C, float* amax_slots, void* stream);
/* (ABI 10, round 6) The two gradient producers of the decoder's backward (autograd of generators/common/blocks.py:70-111) writing the NEXT
 * contraction's operand directly in the bf16 (out_lo == NULL) / bf16x3 (hi + lo) modes, where gradient operands carry no scale: out_hi / out_lo
 * [N][H][W][C] (C % 8 == 0) hold exactly what lp_act_pack(prologue 0) would write for the fp32 result.
 *   lp_adain_relu_bwd_planes: lp_adain_relu_bwd; keep_dx = 0: planes only (dx is scratch: it holds the masked gradient between the passes)
 *   lp_sum2x2_planes:         lp_sum2x2; out = NULL: planes only */
int lp_adain_relu_bwd_planes(const float* dA, const float* x, const float* add, const float* gamma, int ab_stride,
                             const float* mean, const float* rstd, const float* scale, const float* shift,
                             float* dx, float* dgamma, float* dbeta, float* workspace,
                             int N, int H, int W, int C, int upsample, uint16_t* out_hi, uint16_t* out_lo, int keep_dx, void* stream);
int lp_sum2x2_planes(const float* in, float* out, uint16_t* out_hi, uint16_t* out_lo, int N, int H, int W, int C, void* stream);

/* (ABI 11, round 6) Backward of conv3x3 + nn.AvgPool2d(2) [+ the next block's in-place ReLU] (the critic's down blocks, generators/common/blocks.py:76-90,
 * discriminators/no_landmarks.py:52-81) as ONE pass over the pooled gradient dy [N][h][w][C] (C % 8 == 0):
 *   dm = dy * [ymask > 0] (ymask [N][h][w][C] fp32 = the stored relu(pooled) output | NULL: no ReLU behind the pool); dm (fp32) | NULL;
 *   lo_hi / lo_lo [N][h][w][C] | NULL: operand planes of dm (what lp_act_pack of dm writes; operand of the data-gradient launch);
 *   up_hi / up_lo [N][2h][2w][C] | NULL: operand planes of 0.25 * nearest_up2(dm) -- the pool's adjoint, operand of lp_conv16_wgrad -- the same 16-bit values
 *   replicated over each 2 x 2 window.  Replaces torch.where(y > 0, dy, 0), lp_act_pack, lp_avgpool2_bwd (a full-resolution fp32 round trip) and a second
 *   lp_act_pack.  fp16 mode: amax_part / amax_count / amax_stride as in lp_act_pack (partials of dy: lp_amax_partial), scale_out[4] (device) =
 *   {s, 1/s, 4 s, 1/(4 s)}: the lo planes carry scale s, the up planes 4 s (0.25 dm * 4 s = dm * s).  bf16 / bf16x3: no scale, scale_out untouched. */
int lp_pool_grad_pack(const float* dy, const float* ymask, float* dm, uint16_t* lo_hi, uint16_t* lo_lo, uint16_t* up_hi, uint16_t* up_lo,
                      int N, int h, int w, int C, int prec, const float* amax_part, int amax_count, int amax_stride, float* scale_out, void* stream);

/* Generator head (noBottleneck.py:86-88,170-181): t = tanh(z), rgb = t[:3]*0.75+0.5, segm = t[3]*0.5+0.5,
 * fake_rgbs = rgb*segm.  z/t NHWC [N][H][W][4]; fake_rgbs NCHW [N][3][H][W]; fake_segm NCHW [N][1][H][W]. */
int lp_head_fwd(const float* z, float* t, float* fake_rgbs, float* fake_segm, int N, int H, int W, void* stream);
int lp_head_bwd(const float* t, const float* d_rgbs, const float* d_segm, float* dz, int N, int H, int W, float* amax_slots, void* stream);

/* ---- discriminator / perceptual-loss helpers (discriminators/no_landmarks.py:52-108, criterions/common/perceptual_loss.py) ---- */
/* dx = dA * [x > 0]                       (autograd of nn.ReLU, blocks.py:71-73,84) */
int lp_relu_bwd(const float* dA, const float* x, float* dx, long long numel, void* stream);
/* y = AvgPool2d(2)(relu?(x)); x [N][2H][2W][C], y [N][H][W][C]   (nn.AvgPool2d, blocks.py:89-90; perceptual_loss.py:77);
 * out_hi [N][H][W][C]|NULL (C % 8 == 0, prec bf16 | fp16): also the operand planes of y for the conv that follows.
 * relu_in is a flag word (ABI 8): bit 0 = ReLU on the input; bit 1 = ReLU on the OUTPUT, y = relu(pool(x)) -- the in-place ReLU the critic's next
 * ResBlock applies to its input (blocks.py:71-73), fused so that the pooled tensor, its feature-list entry and the next conv's planes come from one
 * pass.  lp_avgpool2_bwd with bit 1: `x` points at that output y [N][H/2][W/2][C] and dy is masked by [y > 0] (bits 0 and 1 are exclusive there). */
int lp_avgpool2_fwd(const float* x, float* y, int N, int H, int W, int C, int relu_in, uint16_t* out_hi, int prec, void* stream);
/* the same on operand planes (ABI 7): x_hi [N][2H][2W][C] -> out_hi [N][H][W][C] (C % 8 == 0, prec bf16 | fp16; fp32 sum, one rounding) --
 * for no-grad chains that keep no fp32 activations: the VGG stacks over the TARGET image (perceptual_loss.py:86-93: `with torch.no_grad()`) */
int lp_avgpool2_fwd16(const uint16_t* x_hi, uint16_t* out_hi, int N, int H, int W, int C, int prec, void* stream);
/* dx [N][H][W][C] = 0.25 * dy[.., y>>1, x>>1, ..] * (relu_in ? [x>0] : 1); H, W = full-resolution dims */
int lp_avgpool2_bwd(const float* dy, const float* x, float* dx, int N, int H, int W, int C, int relu_in, float* amax_slots, void* stream);
/* the same with the ReLU mask read from the operand planes mask_hi [N][H][W][C] of relu(x) (ABI 7: planes-only chains keep no fp32 x) */
int lp_avgpool2_bwd_m16(const float* dy, const uint16_t* mask_hi, float* dx, int N, int H, int W, int C, float* amax_slots, void* stream);
/* Reflection padding of the 3x3 ResBlock convs (ABI 9) -- reference: generators/common/blocks.py:76-88 `padding(1)` = nn.ReflectionPad2d(1) in
 * front of a conv with padding 0 (--gen_padding / --dis_padding 'reflection': generators/vector_pose_unsupervised_segmentation_noBottleneck.py:53-58,
 * discriminators/no_landmarks.py:45-50).  conv(reflect_pad(x)) = conv_zero_pad(x) + B(x), where B adds, at the 2W + 2(H-2) border pixels, the taps that
 * leave the image applied to the mirrored source pixel (-1 -> 1, H -> H-2).  The three calls add B, B^T and the border's share of the weight gradient
 * to the results of lp_conv16_fwd / lp_conv16_fwd (data gradient) / lp_conv16_wgrad with zero padding.  H, W = the conv's resolution (>= 4); x planes
 * [N][H >> upsample][W >> upsample][C8] in operand mode `prec` (x_lo: bf16x3 only) = what the conv consumed; w = W_orig [Cout][Cin][3][3] fp32;
 * alpha = device scalar 1/sigma | NULL.
 *   fwd:   y  [N][H][W][Cout] += alpha * sum_{taps t outside at p} W[:, :, t] . x[mirror(p + t)]
 *   dgrad: dx [N][H][W][Cin]  += alpha * (the transposed sum; gathered per pixel of the ring one pixel inside the border, no atomics), multiplied by
 *          [mask_hi > 0] when mask_hi (planes [N][H][W][mask_c8] of relu(x), the forward's prologue) is given
 *   wgrad: gw [Cout][Cin][3][3] = sum_n sum_{p: p + t outside} dy[n][p][:] (x) x[n][mirror(p + t)][:]   (every element written, centre tap 0): the raw
 *          gradient w.r.t. W/sigma -- lp_sn_grad_apply turns it into the W_orig gradient like the main term (the rule is linear in it).  Thin layers cut
 *          their border pixels into slices (partials in `workspace`, added in a fixed order: deterministic). */
int lp_reflect_border_fwd(const uint16_t* x_hi, const uint16_t* x_lo, int prec, int N, int H, int W, int Cin, int C8, int upsample,
                          const float* w, int Cout, const float* alpha, float* y, void* stream);
int lp_reflect_border_dgrad(const float* dy, int N, int H, int W, int Cout, const float* w, int Cin, const float* alpha,
                            const uint16_t* mask_hi, int mask_c8, float* dx, void* stream);
int lp_reflect_border_wgrad(const uint16_t* x_hi, const uint16_t* x_lo, int prec, int N, int H, int W, int Cin, int C8, int upsample,
                            const float* dy, int Cout, float* gw, float* workspace, void* stream);
/* bytes of `workspace` for that call (0: none needed -- the layer has enough output tiles to fill the chip without slicing its border pixels) */
long long lp_reflect_border_wgrad_workspace_bytes(int N, int H, int W, int Cin, int Cout);
/* L1 taps: partial[lp_l1_partial_blocks()] block sums of |relu?(a) - relu?(b)| (F.l1_loss numerator; featmat.py:17, perceptual_loss.py:107);
 * backward: da = coef * grad_out[0] * sign(relu?(a) - relu?(b)) * (relu_in ? [a>0] : 1)  (+ add [numel]|NULL: the gradient that
 * reaches `a` from its other consumer -- the next conv / pool of the VGG stack -- summed here instead of by an autograd add) */
int lp_l1_partial_blocks(void);
/* out|NULL: a second tiny launch writes out[0] = coef * sum(partial) (fixed order) -- the finished loss term.
 * sign_out [numel] int8|NULL: the forward also leaves sign(relu?(a) - relu?(b)) * (relu_in ? [a > 0] : 1); lp_l1_bwd(sign = that) then
 * reads one byte per element instead of a and b again (a, b may be NULL). */
int lp_l1_fwd(const float* a, const float* b, float* partial, long long numel, int relu_in, float coef, float* out, int8_t* sign_out,
              void* stream);
/* lp_l1_fwd with b given as 16-bit operand planes b_hi [numel] (prec bf16 | fp16; same element order as a) -- ABI 7 */
int lp_l1_fwd_b16(const float* a, const uint16_t* b_hi, int prec, float* partial, long long numel, int relu_in, float coef, float* out,
                  int8_t* sign_out, void* stream);
/* both operands as the 16-bit operand planes of relu(.) (ABI 7; relu_in semantics: the sign pattern carries the [a > 0] mask) */
int lp_l1_fwd_ab16(const uint16_t* a_hi, const uint16_t* b_hi, int prec, float* partial, long long numel, float coef, float* out,
                   int8_t* sign_out, void* stream);
int lp_l1_bwd(const float* a, const float* b, const float* grad_out, float coef, const float* add, float* da, long long numel, int relu_in,
              const int8_t* sign, float* amax_slots, void* stream);

/* Loss reductions.  lp_reduce_dice (criterions/dice.py:20-39): fake [B][Cf][HW], real [B][Cr][HW] NCHW, Cf = 1 (broadcast over real's
 * channels -- the reference's 1-vs-3 channel quirk: overlap and sum r^2 over B x Cr, sum f^2 over B x 1) or Cf = Cr;
 * out[0] = -log(sum 2 f r / (sum f^2 + sum r^2)) * weight; partial = lp_dice_partial_blocks() * 3 floats; sums[2] = {overlap, energy}
 * kept for lp_reduce_dice_bwd: dfake = -weight * grad_out[0] * (2 sum_c r / overlap - 2 f / energy).
 * lp_reduce_hinge (criterions/adversarial.py:34-57, gan_type 'gan'): out[0] = loss_G = -mean(fake_g),
 * out[1] = loss_D = mean(relu(1 - real)) + mean(relu(1 + fake_d)) over B scores; backward writes the non-NULL ones of
 * d_real, d_fake_d (need grad_D), d_fake_g (needs grad_G). */
int lp_dice_partial_blocks(void);
int lp_reduce_dice(const float* fake, const float* real, float* partial, float* out, float* sums, int B, int Cf, int Cr, int HW,
                   float weight, void* stream);
int lp_reduce_dice_bwd(const float* fake, const float* real, const float* sums, const float* grad_out, float* dfake, int B, int Cf,
                       int Cr, int HW, float weight, void* stream);
int lp_reduce_hinge(const float* real, const float* fake_d, const float* fake_g, float* out, int B, void* stream);
int lp_reduce_hinge_bwd(const float* real, const float* fake_d, const float* grad_G, const float* grad_D, float* d_real,
                        float* d_fake_d, float* d_fake_g, int B, void* stream);

/* (ABI 11, round 6) Projection head of the critic (discriminators/no_landmarks.py:100-108): out = relu(out) [N][HW][C] NHWC; pooled = sum over HW;
 * score = linear(pooled) + <pooled, embed>.  lp_proj_score_fwd: pooled [N][C] and dot [N] (|NULL with embed NULL) in one launch -- the linear layer
 * stays lp_linear_fwd on pooled.  lp_proj_score_bwd: d_out[n,p,c] = (g_pooled[n,c] + g_dot[n] embed[n,c]) [out > 0] (|NULL), d_embed[n,c] = g_dot[n]
 * pooled[n,c] (|NULL); g_pooled / g_dot | NULL = zero. */
int lp_proj_score_fwd(const float* out, const float* embed, float* pooled, float* dot, int N, int HW, int C, void* stream);
int lp_proj_score_bwd(const float* out, const float* embed, const float* pooled, const float* g_pooled, const float* g_dot, float* d_out,
                      float* d_embed, int N, int HW, int C, void* stream);
/* (ABI 11, round 6) Input side of the VGG criterions (criterions/common/perceptual_loss.py:72-80,86-93): x NCHW [N][3][HW] in [-1, 1] ->
 * out NHWC [N][HW][3] = ((x + 1) / 2 - mean[c]) / std[c] (mean, std: 3 device floats), the reference's operations in the reference's order;
 * lp_image_prep_bwd: dx NCHW = g NHWC / std[c] / 2. */
int lp_image_prep_fwd(const float* x, const float* mean, const float* stdv, float* out, int N, int HW, void* stream);
int lp_image_prep_bwd(const float* g, const float* stdv, float* dx, int N, int HW, void* stream);

/* ---- fused multi-tensor optimizers + EMA (runners/holycow.py:34-41,99-109; utils/radam.py:29-95; torch.optim.Adam) ----
 * table: DEVICE array of {float* p; const float* g; float* m; float* v; long long n;} (lp_mt_desc_bytes() each), one per
 * parameter tensor; step: DEVICE int64 counter, incremented by the call (graph-replay safe).  kind 0 = RAdam, 1 = Adam.
 * lp_mt_ema: p <- p*alpha + g*(1-alpha) (copy_only=1: p <- g, used for buffers). */
int lp_mt_desc_bytes(void);
int lp_mt_optimizer_step(const void* table, int num_tensors, long long max_numel, long long* step, int kind, float lr,
                         float beta1, float beta2, float eps, void* stream);
int lp_mt_ema(const void* table, int num_tensors, long long max_numel, float alpha, int copy_only, void* stream);

/* ---- batched spectral normalisation (legacy torch.nn.utils.spectral_norm hook; generators/common/blocks.py:76-88) ----
 * table: DEVICE array of {const float* w; float* u; float* v; float* u_out; float* v_out; float* sig_out; int rows; int cols;
 * float* part; int rows; int cols; float eps; int pad;} (lp_sn_desc_bytes() each), one per layer; `part` = scratch of
 * ceil(rows/lp_sn_row_block())*cols + rows + ceil(cols/64) floats.  do_iter=1 (train): v <- normalize(W^T u), u <- normalize(W v) in place;
 * always: u_out/v_out = the vectors used, sig_out = {sigma = u^T W v, 1/sigma}.  Five launches, row-/column-blocked over many workgroups.
 * (dot = scratch of 512 floats: per-block partial sums of <g, w_orig>, no memset needed)
 * lp_sn_grad_apply: g/sigma - (<g, w_orig>/sigma^2) u v^T (autograd of W/sigma with u, v constant), written in place on g, or
 * added to `accum` when that is non-NULL (fused accumulation into the parameter's .grad; g is then left untouched).
 * ndot = 0: dot = scratch of 512 floats, <g, w_orig> is taken here; ndot > 0: dot holds that many partial sums of <g, w_orig>. */
int lp_sn_desc_bytes(void);
int lp_sn_row_block(void);
int lp_sn_power_iter(const void* table, int num_layers, int do_iter, int max_rows, int max_cols, void* stream);
int lp_sn_grad_apply(float* g, const float* w_orig, const float* u, const float* v, const float* sig, float* dot, int ndot,
                     float* accum, int rows, int cols, void* stream);
/* (ABI 8) `count` lp_sn_grad_apply jobs in ceil(count / 40) launches.  host_descs: HOST array of lp_sn_apply_desc_bytes()-sized records
 * {float* g; const float* u, *v, *sig, *dot; float* accum; int ndot, rows, cols, reserved} (device pointers inside; read at call time and passed
 * to the kernel by value, so the launch is hipGraph-capturable without a device table).  Every job accumulates into its OWN accum target and
 * brings its <g, w_orig> partials (ndot >= 1: what lp_conv16_wgrad's reduction left). */
int lp_sn_apply_desc_bytes(void);
int lp_sn_grad_apply_batch(const void* host_descs, int count, void* stream);
/* Label embedding of the projection critic (discriminators/no_landmarks.py:84-86,152), gradient w.r.t. W_orig [N][E] ADDED to grad:
 * grad[label[b]] += rows[b] (b = 0 .. B-1, in order: duplicate labels accumulate deterministically) and grad -= coef * u v^T with
 * coef a device scalar (<G, W_orig> / sigma^2), u [N], v [E] the power-iteration vectors.  label: int64.  E % 4 == 0. */
int lp_sn_embed_grad(float* grad, const float* u, const float* v, const float* coef, const long long* label, const float* rows,
                     int N, int E, int B, void* stream);

#ifdef __cplusplus
}
#endif
#endif

/*
 * dcs_hip.h -- C ABI of libdcs_hip.so: the MI355X (gfx950) kernels behind the
 * doubly-contrastive segmentation train step.
 *
 * Boundary rules:
 *   - plain C: raw DEVICE pointers, explicit shapes, a hipStream_t passed as void*;
 *   - no allocation inside: every workspace is a caller-provided pointer;
 *   - launch-only: never synchronises, never throws; returns 0 or a negative DCS_E_* code;
 *   - thread-safe as long as streams and buffers differ.
 * Activations are NHWC fp32 ("pixel rows of C contiguous channels").  Convolution weights are
 * KRSC fp32 = the physical layout of a PyTorch OIHW tensor in channels_last memory format, so
 * the reference's state_dict shapes are kept while the kernels read K-contiguous rows.
 *
 * Each entry point cites the reference code it replaces (paths relative to the reference root).
 */
#ifndef DCS_HIP_H
#define DCS_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DCS_OK 0
#define DCS_E_ARG (-1)      /* bad shape / alignment / null pointer */
#define DCS_E_LAUNCH (-2)   /* hipGetLastError() != hipSuccess after the launch */
#define DCS_E_UNSUPPORTED (-3)

#define DCS_MAX_TAPS 49

/* ---- library switches -------------------------------------------------------------------------
 * A/B and test switches of the kernels (no reference counterpart).  Their initial values are read from the environment
 * ONCE, when the library is loaded (DCS_BN_NT, DCS_NT_MIN_MB, DCS_X3_BM128, DCS_X3_HALO, DCS_WGRAD_ROLL, DCS_CONV_BK16,
 * DCS_WGRAD_CH32, DCS_CONTRAST_FUSED); no launcher reads the environment.  Names: "bn_nt", "nt_min_mb", "x3_bm128",
 * "x3_halo", "wgrad_roll", "conv_bk16", "wgrad_ch32", "contrast_fused" (csrc/dcs_config.h says what each one does).  Not thread-safe against concurrent launches. */
int dcs_set_option(const char* name, int value);
int dcs_get_option(const char* name, int* value);

/* Gather geometry shared by forward conv, data-gradient and weight-gradient kernels.
 * The GEMM row index m enumerates a sub-grid (n, ty, tx), ty < TY, tx < TX of the DESTINATION
 * tensor: dest pixel = (ty*dsy + dy0, tx*dsx + dx0).  For tap t the SOURCE pixel is
 * (ty*sy + offy[t], tx*sx + offx[t]) (zero outside [0,SH)x[0,SW)); its K channels are multiplied
 * with weight row segment  w[co*wstride + wofs[t] .. +K).
 *   forward conv  : sub-grid = whole output, sy = stride, off = r - pad, wofs = (r*S+s)*Cin
 *   data gradient : one launch per input parity class, sy = 1, off = (py+pad-r)/stride
 *   stem (7x7/2 on NHWC4): tap = kernel row r, the 7 kernel columns x 4 channels form one 32-wide
 *                   K chunk (28 valid), mode flag stem = 1.                                      */
typedef struct DcsConvGeom {
  int32_t N, SH, SW;          /* source tensor [N,SH,SW,*]                      */
  int32_t DH, DW;             /* destination tensor [N,DH,DW,*]                 */
  int32_t TY, TX;             /* sub-grid extent; M = N*TY*TX                   */
  int32_t sy, sx;             /* source step per sub-grid index                 */
  int32_t dsy, dsx, dy0, dx0; /* destination step / origin                      */
  int32_t K;                  /* channels per tap taken from the source         */
  int32_t Cout;               /* GEMM N                                         */
  int32_t ntaps;
  int32_t wstride;            /* floats between consecutive weight rows         */
  int32_t src_cstride;        /* floats between consecutive source pixels       */
  int32_t dst_cstride;        /* floats between consecutive destination pixels  */
  int32_t stem;               /* 1: NHWC4 stem mode (see above)                 */
  int16_t offy[DCS_MAX_TAPS];
  int16_t offx[DCS_MAX_TAPS];
  int32_t wofs[DCS_MAX_TAPS];
} DcsConvGeom;

/* ---- convolution as implicit GEMM on v_mfma_f32_32x32x2_f32 --------------------------------
 * replaces nn.Conv2d forward (network/backbone/resnet_pyramid.py:23-25,:139,:110-112;
 * network/utils.py:46-47) and the data half of aten::convolution_backward.
 * dst[m, co] (+)= bias[co] + sum_t sum_k src[gather(m,t), k] * wgt[co, wofs[t]+k]
 * stats (nullable): [ceil(M/DCS_CONV_BM)][2][Cout] per-row-tile sums / sums of squares of the values written
 * (the batch statistics of the BatchNorm that follows); reduce them with dcs_colsum_final.             */
#define DCS_CONV_BM 128
int dcs_conv_gather(const float* src, const float* wgt, const float* bias, float* dst,
                    const DcsConvGeom* geom, int accumulate, float* stats, void* stream);

/* Data gradient with the BatchNorm-backward reductions in its epilogue: as dcs_conv_gather (no bias), and part
 * [ceil(M/DCS_CONV_BM)][2][Cout] receives per row tile sum(gm) and sum(gm * xhat), gm = (final dst value, after the
 * optional accumulate) * ReLU mask, xhat = (bn_y - mean) * invstd -- the sums dcs_colsum_partial(mode 1) would take in a
 * pass of its own.  bn_y (and bn_mask, nullable: mask = bn_mask > 0; else relu ? bn_y*scale+shift > 0 : 1) have dst's
 * layout; relu = 2: bn_mask points to the BYTE mask dcs_bn_act wrote for that tensor (four bits per float4; 1/16 of the bytes);
 * relu | 4: dst receives the MASKED gradient gm instead of the raw one (a residual block's incoming gradient is only ever
 * used behind its ReLU: the BatchNorm backward that follows then has no gm to write); dst must be dense (dst_cstride == Cout, Cout % 4 == 0).  Stride-2 data gradients are several launches (one
 * per input parity class), each with its own rows of part; dcs_colsum_final sums them all. */
int dcs_conv_gather_bnbwd(const float* src, const float* wgt, float* dst, const DcsConvGeom* geom, int accumulate,
                          const float* bn_y, const float* bn_mask, const float* bn, int relu, float* part, void* stream);

/* Split-K variant of dcs_conv_gather for launches with few output tiles (deep layers of small inputs: a handful of
 * blocks behind a 100+-chunk serial loop leaves most CUs idle): grid.y = nsplit, split s reduces its share of the
 * (tap, channel-chunk) range and writes its partial output to slab + s * slab_stride (same addressing as dst;
 * slab_stride >= N*DH*DW*dst_cstride elements).  dcs_reduce_slab then sums the slabs in fixed order (deterministic),
 * optionally accumulating into dst.  No bias, no fused statistics. */
int dcs_conv_gather_split(const float* src, const float* wgt, float* slab, const DcsConvGeom* geom, int nsplit,
                          int64_t slab_stride, void* stream);

/* Weight gradient, split over pixel ranges.  slab[split][co][wstride] receives partial sums for
 * split = split0 .. split0+nsplit-1 (every element of those slab slices is written).
 * replaces the weight half of aten::convolution_backward.                                      */
int dcs_conv_wgrad(const float* src, const float* dy, float* slab, const DcsConvGeom* geom,
                   int dy_cstride, int split0, int nsplit, void* stream);

/* Convolution / weight gradient of the ACTIVATED source: the operand is relu(src * scale[k] + shift[k]) (zero in the
 * padding, exactly as a convolution of the materialised activation), applied on the way from the staging registers into
 * LDS, so the BatchNorm + ReLU output between two convolutions (network/backbone/resnet_pyramid.py:62-66 conv2 of a
 * BasicBlock; network/utils.py:44-47 _BNReluConv) is never written to or read from HBM.  pro = [scale(K), shift(K)], the
 * head of a dcs_bn_finalize record; K <= DCS_PRO_MAXK, not for the stem.  dcs_conv_gather_pro: nsplit > 1 is the
 * split-K form (dst = slab, see dcs_conv_gather_split; no bias / stats / accumulate). */
#define DCS_PRO_MAXK 512
int dcs_conv_gather_pro(const float* src, const float* wgt, const float* bias, float* dst, const DcsConvGeom* geom,
                        int accumulate, float* stats, const float* pro, int nsplit, int64_t slab_stride, void* stream);
int dcs_conv_wgrad_pro(const float* src, const float* dy, float* slab, const DcsConvGeom* geom, int dy_cstride,
                       int split0, int nsplit, const float* pro, void* stream);

/* ---- the same gather on the bf16 matrix cores, fp32 operands as three bf16 pieces (csrc/conv_split.hip) --------------
 * x = x1 + x2 + x3 exactly (3 x 8 significand bits); a*b is accumulated in fp32 from the six piece products >= 2^-16|ab|,
 * the dropped three are <= 2^-24 |ab| each (below one fp32 rounding): fp32-class error at 6/16 of the fp32-MFMA time.
 * dcs_split_weight: w [rows][wstride] (wstride % 16 == 0) -> out [rows][wstride/16][3][16] bf16 (6 bytes / element).
 * dcs_conv_gather_x3 = dcs_conv_gather / _pro / _bnbwd / _split in one entry (pro, bn_y nullable; nsplit > 1: dst = slab,
 * no bias / stats / accumulate) for geometries with Cout > 32, K % 16 == 0, wofs % 16 == 0, and for the stem in its 14-tap
 * form (a tap = a filter half-row of 4 px x 4 ch = one 16-float chunk; Cout 64, wstride 224, no prologue / K split);
 * everything else returns DCS_E_UNSUPPORTED and stays on dcs_conv_gather.  Dense 3x3 / stride 1 / pad 1 geometries on
 * maps with TX % 32 == 0 and enough tiles run the variant that keeps the input halo resident in LDS (DCS_X3_HALO=0: never,
 * =2: whenever the geometry allows).  wsplit = dcs_split_weight of the fp32 weight the fp32
 * entry would take.  Non-finite inputs give NaN (inf - inf in the split), not inf. */
int dcs_split_weight(const float* w, void* out, int64_t rows, int wstride, void* stream);
/* accumulate | DCS_ACC_FP16X2 (defined below): the fp16 two-piece form, as for dcs_conv3x3_x3w -- wsplit from
 * dcs_split_weight_h2 ([rows][wstride/16][2][16] fp16 of w * 2^10), src_max as there (nullable: src scaled by 2^2). */
int dcs_split_weight_h2(const float* w, void* out, int64_t rows, int wstride, void* stream);
int dcs_conv_gather_x3(const float* src, const void* wsplit, const float* bias, float* dst, const DcsConvGeom* geom,
                       int accumulate, float* stats, const float* pro, const float* bn_y, const float* bn_mask,
                       const float* bn, int relu, int nsplit, int64_t slab_stride, const uint32_t* src_max, void* stream);

/* Dense 3x3 / stride 1 / pad 1 geometries (TX % 32 == 0, TY % 8 (Cout <= 64) or % 4 == 0, K % 32 == 0), weights straight
 * from global memory into the MFMA fragments: dcs_split_weight_frag lays them out fragment-major (unit
 * (((c*J + j)*3 + p)*2 + h)*32 + r of 16 bytes = piece p of row 32j + r, channels 16c + 8h .. +7, J = ceil(rows / 32)),
 * followed by a sign-flipped copy (2 x wstride/16 x J x 3 KiB in all); dcs_conv3x3_x3w is dcs_conv_gather_x3 (one K
 * split) on that layout.  Other geometries: DCS_E_UNSUPPORTED.
 * accumulate | DCS_ACC_FP16X2 (forward convolutions): the operands are split into TWO fp16 pieces instead of three bf16
 * ones -- three MFMAs per product instead of six; activations scaled by 2^2, weights by 2^10 (exact), i.e. defined for
 * |x| < 16384 and |w| < 64 (beyond: inf, like any fp16 conversion), same fp32-class error (csrc/conv_split.hip, split2h_quad).
 * wfrag must then come from dcs_split_weight_frag_h2 (unit (((c*J + j)*2 + p)*2 + h)*32 + r; 2 x wstride/16 x J x 2 KiB).
 * src_max (nullable, DCS_ACC_FP16X2 only): device word holding the bit pattern of max |src| (as dcs_bn_bwd_apply's
 * dy_maxabs leaves it).  The kernel then scales src by the power of two that puts that maximum into [2^13, 2^14) instead of
 * 2^2: data gradients (src = dy, magnitudes of 1e-3 .. 1e-10) use the same kernel with an exact, per-tensor scale.
 * Stem (network/backbone/resnet_pyramid.py:110-112, 7x7 / stride 2 / pad 3 on the NHWC4 image): with the 14-tap stem geometry,
 * accumulate | DCS_ACC_FP16X2, wfrag = dcs_split_weight_frag_h2(packed [64][7][8][4] weight, 64, 224), no bias / pro / bn_*
 * / src_max, and an output map of whole 8 x 32-pixel tiles the same entry runs the stem forward with its input patch
 * resident in LDS (stem7_h2_kernel); other stem launches: DCS_E_UNSUPPORTED (use dcs_conv_gather_x3). */
#define DCS_ACC_FP16X2 16
/* with DCS_ACC_FP16X2 on dcs_conv_gather_x3 (not the stem): wsplit is the FRAGMENT-major image of dcs_split_weight_frag_h2 and
 * the kernel reads its weight fragments straight from global memory instead of staging them through LDS (round 3: the
 * LDS-staged fp16 form was LDS-bandwidth bound) -- same products in the same order, bitwise the same results. */
#define DCS_ACC_WFRAG 32
int dcs_split_weight_frag(const float* w, void* out, int64_t rows, int wstride, void* stream);
int dcs_split_weight_frag_h2(const float* w, void* out, int64_t rows, int wstride, void* stream);
int dcs_conv3x3_x3w(const float* src, const void* wfrag, const float* bias, float* dst, const DcsConvGeom* geom,
                    int accumulate, float* stats, const float* pro, const float* bn_y, const float* bn_mask,
                    const float* bn, int relu, const uint32_t* src_max, void* stream);

/* ---- level batching: up to DCS_MULTI_MAX launches of the entries above as ONE grid ---------------------------------------
 * The three pyramid levels of a layer share every weight (network/backbone/resnet_pyramid.py:318-341: the same modules are
 * applied to the image at scales 1, 1/2, 1/4) and therefore run the same kernel on maps of different extent.  A multi
 * entry takes the argument lists of n single launches (one struct each, fields = the arguments of the single entry),
 * validates every one exactly as the single entry would (nothing is launched unless all are valid; the first failing
 * code is returned) and launches those that select the same kernel as one grid whose block ranges are assigned to the
 * sub-launches; sub-launches that select different kernels go out separately.  A block executes the code of the single
 * launch with its sub-launch's arguments, so results are BITWISE those of n single launches in any order; the
 * sub-launches must not write overlapping memory (the levels' tensors never do). */
#define DCS_MULTI_MAX 3
typedef struct DcsGatherLaunch {
  const float* src; const void* wgt; const float* bias; float* dst; const DcsConvGeom* geom;
  float* stats; const float* pro; const float* bn_y; const float* bn_mask; const float* bn;
  int64_t slab_stride;
  int32_t accumulate, relu, nsplit;
  const uint32_t* src_max;   /* with DCS_ACC_FP16X2 only (else null): see dcs_conv3x3_x3w */
} DcsGatherLaunch;
int dcs_conv_gather_x3_multi(const DcsGatherLaunch* launches, int n, void* stream);   /* n x dcs_conv_gather_x3 */
int dcs_conv3x3_x3w_multi(const DcsGatherLaunch* launches, int n, void* stream);      /* n x dcs_conv3x3_x3w (nsplit 1) */

/* dcs_conv_wgrad / dcs_conv_wgrad_pro (pro nullable) on the bf16 matrix cores, operands as three bf16 pieces; slabs of
 * odd split index carry the hardware's rounding bias with the opposite sign, so an EVEN nsplit cancels it in
 * dcs_reduce_slab.  Cout % 4 == 0; the stem in its seven-tap form with TX % 16 == 0, Cout 64, wstride 224, no prologue
 * (else DCS_E_UNSUPPORTED).
 * dy_max (nullable): the device word dcs_bn_bwd_apply left with the bits of max |dy|.  With it, dense 3x3 / stride 1
 * geometries that take the rolling-window kernel run on TWO fp16 pieces per operand (three MFMAs per product; the input
 * scaled by 2^2, dy by the power of two that puts its maximum into [2^13, 2^14), both exact); other geometries ignore it. */
int dcs_conv_wgrad_x3(const float* src, const float* dy, float* slab, const DcsConvGeom* geom, int dy_cstride,
                      int split0, int nsplit, const float* pro, const uint32_t* dy_max, void* stream);
typedef struct DcsWgradLaunch {
  const float* src; const float* dy; float* slab; const DcsConvGeom* geom; const float* pro;
  int32_t dy_cstride, split0, nsplit;
  const uint32_t* dy_max;
} DcsWgradLaunch;
int dcs_conv_wgrad_x3_multi(const DcsWgradLaunch* launches, int n, void* stream);     /* n x dcs_conv_wgrad_x3, see above */


/* dw[o(i)] = (accumulate ? dw[o(i)] : 0) + sum_s slab[s][i], fixed order (deterministic).
 * row_len == 0: o(i) = i.  row_len > 0: the slab holds compact rows of row_len floats that land at stride
 * dst_stride in dw (gradient of a channel slice of a wider weight, "virtual concat" convolutions). */
int dcs_reduce_slab(const float* slab, float* dw, int64_t n, int nsplit, int accumulate, int row_len,
                    int dst_stride, void* stream);

/* [Cout][R][S][cin_off .. cin_off+Cin) of rows with cin_total channels -> [Cin][R][S][Cout]
 * (weights for the data-gradient GEMM; cin_total = Cin, cin_off = 0 for a whole weight). */
int dcs_pack_dgrad_weight(const float* w_krsc, float* w_crsk, int Cout, int R, int S, int Cin, int cin_total,
                          int cin_off, void* stream);
/* stem: [64][7][7][3] <-> [64][7][8][4] zero padded (dir 0: pack, 1: unpack). */
int dcs_pack_stem_weight(const float* in, float* out, int Cout, int dir, void* stream);
/* out[c][r] = in[r][c]  (in: [R][C]) */
int dcs_transpose(const float* in, float* out, int R, int C, void* stream);

/* ---- per-channel reductions and BatchNorm ----------------------------------------------------
 * replaces nn.BatchNorm2d train/eval forward+backward (resnet_pyramid.py:61,:65,:159-161;
 * network/utils.py:36-41).                                                                     */
/* partial[b][g][2][C]: per group g of rows, sum and sum of squares of x[b][rows][C].
 * mode 0: (sum x, sum x^2).  mode 1: BN backward sums (sum gm, sum gm*xhat) where
 * gm = g * mask, mask = (masksrc ? masksrc > 0 : (relu ? y*scale+shift > 0 : 1)),
 * xhat = (y - mean) * invstd.  bn = [scale, shift, mean, invstd] x C (may be null in mode 0).  */
int dcs_colsum_partial(const float* x, const float* y, const float* masksrc, const float* bn,
                       float* partial, int B, int64_t rows, int C, int cstride, int groups,
                       int mode, int relu, void* stream);
/* out[b][2][C] = sum over groups, accumulated in double, times scale.
 * moments_count > 0 (needs scale == 1): out[b][0][c] = mean = sum/count, out[b][1][c] = biased variance
 * = sumsq/count - mean^2, both formed in double before the single rounding to fp32 (dcs_bn_finalize training = 2). */
int dcs_colsum_final(const float* partial, float* out, int B, int groups, int C, float scale, double moments_count,
                     void* stream);
/* From sums[2][C] over `count` rows: bn[0..4C) = scale, shift, mean, invstd; updates running
 * mean/var `repeats` times (momentum, unbiased variance) when running_mean != null.
 * training = 0: ignores sums and derives scale/shift from the running statistics.
 * training = 2: sums holds (mean, biased variance) from dcs_colsum_final(moments_count = count).  */
int dcs_bn_finalize(const float* sums, const float* gamma, const float* beta, float* running_mean,
                    float* running_var, float* bn, int C, double count, float eps, float momentum,
                    int repeats, int training, void* stream);
/* Re-applies the running-stat EMA with saved batch statistics (activation-checkpoint recompute
 * side effect): bn holds mean/invstd; count rows. */
int dcs_bn_ema_again(const float* bn, float* running_mean, float* running_var, int C, double count,
                     float eps, float momentum, void* stream);

/* z = act(y*scale+shift [+ r | + r*scale2+shift2]); act = relu if relu.  bn2 null -> identity r.
 * mask8 (nullable, rows * C / 4 bytes): byte i receives the four bits (z > 0) of float4 i -- the ReLU mask of a residual
 * block's output (resnet_pyramid.py:38-60, `relu(out + residual)`), which dcs_bn_bwd_apply (mask8) and the
 * BatchNorm-backward epilogue of a data gradient (relu = 2) then read instead of the 16 bytes of z. */
int dcs_bn_act(const float* y, const float* bn, const float* r, const float* bn2, float* z,
               int64_t rows, int C, int relu, uint8_t* mask8, void* stream);
/* BN backward apply: gm = g*mask; dy (+)= gamma*invstd*(gm - s0/cnt - xhat*s1/cnt);
 * optional gm_out (+)= gm.  sums = [2][C] from dcs_colsum_* mode 1.  dgamma/dbeta (+)= s1/s0.
 * training = 0 (eval-mode BN, running statistics): dy = gamma*invstd*gm.
 * dy_maxabs (nullable): device word, zero before the call; receives the bit pattern of max |dy| over the written tensor
 * (integer atomicMax of non-negative floats: order-independent, deterministic) -- the per-tensor scale of the fp16
 * two-piece convolution kernels that consume dy (dcs_conv3x3_x3w src_max, dcs_conv_wgrad_x3 dy_max). */
/* out (zero before the call) receives the bit pattern of max |x| over n floats (n % 4 == 0), like dy_maxabs below: the
 * per-tensor scale of the fp16 two-piece convolution kernels for a gradient that no BatchNorm backward produced. */
int dcs_maxabs(const float* x, int64_t n, uint32_t* out, void* stream);
int dcs_bn_bwd_apply(const float* g, const float* y, const float* masksrc, const float* bn,
                     const float* gamma, const float* sums, float* dy, float* gm_out,
                     float* dgamma, float* dbeta, int64_t rows, int C, int relu, int acc_dy,
                     int acc_gm, int acc_param, int training, uint32_t* dy_maxabs, const uint8_t* mask8,
                     void* stream);   /* mask8 (nullable; then masksrc must be null): the byte mask dcs_bn_act wrote */

/* ---- pyramid / pooling / resize -------------------------------------------------------------*/
/* resnet_pyramid.py:296-314: (x-mean)/std then bicubic 1/2 and 1/4 (A=-0.75, no antialias).
 * img NCHW [N,3,H,W] raw; out0/1/2 NHWC4 (4th channel 0) at H, H/2, H/4 (floor). */
int dcs_normalize_pyramid(const float* img, float* out0, float* out1, float* out2, int N, int H, int W,
                          const float* mean3, const float* std3, void* stream);
/* resnet_pyramid.py:322-325: maxpool3x3/2 pad1 of relu(y*scale+shift).  idx: uint8 argmax 0..8. */
int dcs_bn_relu_maxpool(const float* y, const float* bn, float* out, uint8_t* idx, int N, int H, int W,
                        int C, void* stream);
/* gz[n,iy,ix,c] = sum of g[outputs whose argmax is (iy,ix)]  (dense, y resolution). */
int dcs_maxpool_bwd(const float* g, const uint8_t* idx, float* gz, int N, int H, int W, int C, void* stream);
/* Backward of bn1 -> relu -> maxpool (resnet_pyramid.py:322-325) WITHOUT the dense pooled-gradient tensor: both passes
 * gather the gradient of a 2x2 pixel quad from its four pooling windows (g, idx at [N,OH,OW,C]) on the fly.
 * partial: [groups][2][C] (sum of masked gradient, sum of gradient * xhat) -> dcs_colsum_final -> sums [2][C];
 * apply: dy [N,H,W,C] = BatchNorm input gradient, dgamma/dbeta written or accumulated.  y = saved conv output.
 * dy_maxabs (may be null): as in dcs_bn_bwd_apply -- the word is raised to the bits of max |dy| (the fp16 two-piece stem
 * weight gradient scales dy by it). */
int dcs_bn_pool_bwd_partial(const float* g, const uint8_t* idx, const float* y, const float* bn, float* partial,
                            int N, int H, int W, int C, int groups, void* stream);
/* dcs_bn_pool_bwd_partial from the POOLED tensors: z [N,OH,OW,C] = the pooled output dcs_bn_relu_maxpool wrote.  A pooled
 * gradient lands on its window's argmax, whose ReLU is open iff z > 0, and there y = (z - shift) / scale: the two sums
 * need g and z only (1/4 of the un-pooled map each); channels with |gamma| < 0.05 read y at the argmax position (idx).
 * Same partial layout; results equal dcs_bn_pool_bwd_partial's to fp32 rounding of xhat (not bitwise). */
int dcs_bn_pool_bwd_partial_pooled(const float* g, const float* z, const uint8_t* idx, const float* y, const float* bn,
                                   float* partial, int N, int H, int W, int C, int groups, void* stream);
int dcs_bn_pool_bwd_apply(const float* g, const uint8_t* idx, const float* y, const float* bn, const float* gamma,
                          const float* sums, float* dy, float* dgamma, float* dbeta, int N, int H, int W, int C,
                          int acc_param, int training, uint32_t* dy_maxabs, void* stream);
/* network/utils.py:92-102: t = bilinear(x -> [OH,OW], align_corners=False) + ((s0+s1)+s2). */
int dcs_upsample_add(const float* x, const float* s0, const float* s1, const float* s2, float* t,
                     int N, int IH, int IW, int OH, int OW, int C, void* stream);
/* dcs_upsample_add that also reduces the batch statistics of the BatchNorm consuming t (network/utils.py:36-41): partial
 * [groups][2][C] = per-block sum / sum of squares of the values written (double accumulators, rounded once per block);
 * groups = the grid (<= 2048, groups * 256 <= N*OH*OW*C/4 + 255), C <= 1024 with 256 % (C/4) == 0; reduce with
 * dcs_colsum_final(partial, out, 1, groups, C, 1, count). */
int dcs_upsample_add_stats(const float* x, const float* s0, const float* s1, const float* s2, float* t, float* partial,
                           int groups, int N, int IH, int IW, int OH, int OW, int C, void* stream);
/* adjoint of the bilinear part: gx[n,iy,ix,c] (+)= sum_o w(o,i) g[n,oy,ox,c].  maxabs (may be null): raised to the bits of
 * max |gx| like dcs_bn_bwd_apply's dy_maxabs (the decoder's next blend convolution scales its gradient by it). */
int dcs_upsample_bwd(const float* g, float* gx, int N, int IH, int IW, int OH, int OW, int C,
                     int accumulate, uint32_t* maxabs, void* stream);
/* network/utils.py:8 on the logits: x NHWC [N,IH,IW,cs] (first C channels) -> out NCHW [N,C,OH,OW]. */
int dcs_upsample_to_nchw(const float* x, float* out, int N, int IH, int IW, int cs, int C, int OH, int OW,
                         void* stream);
/* adjoint: g NCHW [N,C,OH,OW] * gscale -> gx NHWC [N,IH,IW,cs] (channels >= C zeroed).
 * tmp: optional scratch of N*C*OH*IW floats; with it (and C <= 32) the adjoint runs as two separable passes
 * (fold X, then fold Y + NHWC transpose) that read g exactly once. */
int dcs_upsample_to_nchw_bwd(const float* g, const float* gscale, float* gx, float* tmp, int N, int IH, int IW,
                             int cs, int C, int OH, int OW, void* stream);

/* Fused logits upsampling + log-softmax + focal / CE loss + gradient (network/utils.py:8, utils/loss.py:41-42,:60-73):
 * logits_lr [N,ih,iw,cs] low-resolution NHWC logits (C <= 20 <= cs used); target / ldw at the full resolution
 * H = F*ih, W = F*iw (F in {2,4}, else DCS_E_UNSUPPORTED: use dcs_upsample_to_nchw + dcs_seg_loss).  The bilinear
 * upsampling (align_corners=False) is evaluated on the fly with the weights of dcs_upsample_to_nchw; grad_lr
 * [N,ih,iw,cs] = d(sum loss)/d(logits_lr) (the adjoint of the upsampling: per-cell corner sums added per low-resolution
 * pixel in fixed order); partial [blocks][2], blocks = dcs_seg_loss_fused_blocks(N, ih, iw), -> dcs_seg_loss_final.
 * Modes as dcs_seg_loss; target is rewritten in place (255 -> 0) unless mode == 4. */
int dcs_seg_loss_fused(const float* logits_lr, int cs, int64_t* target, const float* ldw, const float* cw,
                       float* grad_lr, float* partial, int N, int C, int ih, int iw, int H, int W, int mode, float gamma,
                       int ignore, int blocks, void* stream);
/* number of blocks (= rows of `partial`) of dcs_seg_loss_fused for this geometry (> 0), or a negative DCS_E_* code */
int dcs_seg_loss_fused_blocks(int N, int ih, int iw);
/* ---- segmentation losses (utils/loss.py:39-80; nn.CrossEntropyLoss) --------------------------
 * logits NCHW [N,C,H,W]; target int64 [N,H,W]; ldw [N,H,W]; cw [C].
 * mode 0: boundary-aware focal (w*alpha), 1: plain_focal, 2: no_class_weights, 3: no_EDT,
 *      4: cross entropy with ignore_index (mean over non-ignored).
 * Focal modes rewrite target==ignore -> 0 in place (loss.py:43).  grad receives the UNSCALED
 * d(sum)/d(logit); partial[blocks][2] = (sum loss, count).  out[0]=loss, out[1]=count after
 * dcs_seg_loss_final, which also leaves 1/count (or 0) in out[2] for the backward scale.      */
int dcs_seg_loss(const float* logits, int64_t* target, const float* ldw, const float* cw, float* grad,
                 float* partial, int N, int C, int H, int W, int mode, float gamma, int ignore,
                 int blocks, void* stream);
int dcs_seg_loss_final(const float* partial, float* out, int blocks, void* stream);
/* x[i] *= a[0]*b[0] (device scalars; b may be null). */
int dcs_scale_inplace(float* x, int64_t n, const float* a, const float* b, void* stream);

/* ---- hard-anchor sampling (utils/loss.py:264-337, :396-410) ----------------------------------*/
/* pred[n,p] = argmax_c logits[n,p,c] (first max), lab[n,p] = labels[n, floor(y*H/h), floor(x*W/w)];
 * key[n,p] = 255 if lab is ignore or out of [0,C), else lab*2 + (pred==lab ? 1 : 0).
 * hist[n][chunk][2C] counts keys per chunk of `chunk` pixels (chunk % 256 == 0).            */
int dcs_anchor_keys(const float* logits, int cs, int C, const int64_t* labels, int N, int h, int w,
                    int H, int W, int ignore, uint8_t* key, int32_t* hist, int chunk, void* stream);
/* req[q] = {n, key, rank}; out[q] = flat pixel index of the rank-th pixel with that key in image n
 * (ascending index order = nonzero() order); -1 if absent.  hist as produced above. */
int dcs_anchor_select(const uint8_t* key, const int32_t* hist, const int32_t* req, int32_t* out, int Q,
                      int N, int HW, int C, int chunk, void* stream);
/* HOST function (no device work, no stream): the sampler's plan between dcs_anchor_keys and dcs_anchor_select -- which
 * (image, class, hard|easy, rank) pixels become anchors (utils/loss.py:264-337).  counts [B][C][2] = pixels per (image,
 * class, hard | easy) (the histogram of dcs_anchor_keys summed over chunks).  mt_state [624] / *mt_pos: the mt19937 of
 * torch's default CPU generator (words, index of the next word; 624 = block exhausted), advanced in place exactly as the
 * reference's torch.randperm(num_hard), torch.randperm(num_easy) per kept class advance it (a CPU randperm(n) is a forward
 * Fisher-Yates shuffle on n - 1 32-bit draws; only the first n_view entries are ever used, the other draws are skipped by
 * regenerating state blocks).  Returns T >= 0 = number of kept (image, class) pairs, with *n_view, req [T * n_view][3] =
 * (image, 2 * class + is_easy, rank) in (t, view) order, cls [T], img [T] (capacity max_samples rows each); 0 = no class
 * qualifies; DCS_E_UNSUPPORTED (generator untouched) where the reference's general path must decide (n_view = 0, or a class
 * the reference raises on). */
int dcs_sampler_plan(const int64_t* counts, int B, int C, int max_samples, int max_views, uint32_t* mt_state, int* mt_pos,
                     int32_t* req, int32_t* cls, int32_t* img, int* n_view);

/* X[a][0..C) = feat[(img[a]*HW + pix[a]) * C ..]  and the adjoint gfeat[...] += gX[a]. */
int dcs_gather_rows(const float* feat, const int32_t* rowidx, float* X, int A, int C, void* stream);
int dcs_scatter_add_rows(const float* gX, const int32_t* rowidx, float* gfeat, int A, int C, void* stream);

/* ---- validation: class-id argmax + confusion matrix (trainer.py:349, metrics/stream_metrics.py:330-342) --------
 * conf[n][gt][pred] (uint64, one C x C matrix per image, += ) over pixels with 0 <= gt < C; pred = argmax_c (first max).
 * lowres_h == 0: logits is the full-resolution NCHW tensor [N,C,H,W].
 * lowres_h  > 0: logits is the low-resolution NHWC tensor [N,lowres_h,lowres_w,cs] and is bilinearly upsampled
 *               (align_corners=False) to HxW on the fly -- the 2.55 GB full-resolution logits never exist.
 * pred_out (nullable): uint8 class ids [N,H,W]. */
int dcs_confusion(const float* logits, const int64_t* labels, uint8_t* pred_out, uint64_t* conf, int N, int C, int H,
                  int W, int lowres_h, int lowres_w, int cs, void* stream);

/* Rows of a bilinear upsampling (align_corners=False) of feat [N,IH,IW,C] to [N,OH,OW] without materialising it
 * (network/utils.py:190 + utils/loss.py:391-415): X[a] = upsampled pixel rowidx[a]; the adjoint adds the four weighted
 * taps of every row into gfeat [N,IH,IW,C] in row order (deterministic). */
int dcs_gather_rows_bilinear(const float* feat, const int32_t* rowidx, float* X, int A, int C, int N, int IH, int IW,
                             int OH, int OW, void* stream);
int dcs_scatter_rows_bilinear(const float* gX, const int32_t* rowidx, float* gfeat, int A, int C, int N, int IH, int IW,
                              int OH, int OW, void* stream);
/* ---- fused similarity / InfoNCE loss (utils/loss.py:339-389 PixelContrastLoss._contrastive = mode 0,
 *      utils/loss.py:175-204 SupConLoss = mode 1), forward AND backward in one call -------------------------------
 * X [A][ldx] anchors (C <= ldx channels used, ldx % 4 == 0), y[i*ldy] float labels; y < 0 marks a padding row (fixed-
 * shape all-gather of the data-parallel step): it contributes to no maximum, norm, denominator, loss or gradient.
 * mask (nullable, mode 1 only): explicit [mask_b][mask_b] positive weights tiled over the views like
 * mask.repeat(anchor_count, contrast_count) (utils/loss.py:148-159, :183); otherwise positives = equal labels, j != i.
 * loss[0] = mean over the valid rows.  Exactly one of:
 *   dX [A][lddx] = d loss / d X = (G + G^T) X            (C <= 128), or
 *   gsym [A][ldg] = G + G^T, G_ij = d loss / d S_ij      (any C; the caller finishes dX with one GEMM).
 * ws: scratch of dcs_contrast_fused_ws(A, C) floats.  A <= 1024: the S strips stay in LDS (statistics, then the gradient
 * sweep); larger A: S is computed once into ws (upper-triangular MFMA tiles), one block per row derives the row records,
 * the gradient product forms G + G^T on the fly (csrc/contrast_large.h).
 * sync (nullable): two uint32 in device memory that are ZERO on entry and are handed back as zeros.  With it the A <= 1024
 * family runs as ONE launch (a grid-wide barrier on that counter between the two phases, <= 64 resident blocks; the gradient
 * sweep then reads the S strip back from LDS); without it, as two launches.  One call at a time per sync word.
 * Rows without positives give NaN like the reference. */
int dcs_contrast_fused_ws(int A, int C, int64_t* floats);
int dcs_contrast_fused(const float* X, int ldx, const float* y, int ldy, const float* mask, int mask_b, int A, int C,
                       int mode, float inv_temp, float* loss, float* dX, int lddx, float* gsym, int ldg, float* ws,
                       int64_t ws_floats, uint32_t* sync, void* stream);

/* ---- contrastive rows, unfused form (kept for A/B measurements in bench.py; not used by the product path) --------
 * S [A,ld] = C C^T (from dcs_conv_gather in 1x1 mode), scaled by inv_temp = 1/T on read.  For every row i < A:
 *   max-subtract, L2-normalise, masked exp/log reductions; loss_row[i]; G[i][j] = d(mean loss)/dS_ij
 * (w.r.t. the unscaled S, already divided by the number of rows A).  labels: float [A].  mode 0: pixel contrast
 * (denominator exp(L_ij)+neg_i), mode 1: SupCon/SimCLR (denominator sum_{k!=i}).  `selfmask` as the
 * reference: positives exclude j == i.  Rows with no positives give NaN like the reference.     */
int dcs_contrast_rows(const float* S, const float* labels, float* loss_row, float* G, int A, int ld,
                      int mode, float inv_temp, void* stream);
/* Gs[i][j] = G[i][j] + G[j][i]  (A x ld, zero padded to ld). */
int dcs_symmetrize(const float* G, float* Gs, int A, int ld, void* stream);
/* out[0] = scale * sum x[0..n) in double (single block, deterministic). */
int dcs_sum_scalar(const float* x, float* out, int n, float scale, void* stream);

/* ---- dropout (nn.Dropout(0.1) of network/_deeplab.py:164) ------------------------------------------
 * out = x * keep / (1-p).  noise != null: keep = noise[i] (0/1 floats supplied by the caller, e.g. drawn from the
 * CPU generator like the reference); else keep is a counter-based hash of (seed, i).  mask_out (uint8, nullable
 * when noise is given) receives keep for dcs_dropout_bwd: out = mask ? g / (1-p) : 0. */
int dcs_dropout(const float* x, const float* noise, uint8_t* mask_out, float* out, int64_t n, float p,
                uint32_t seed, void* stream);
int dcs_dropout_bwd(const float* g, const uint8_t* mask, float* out, int64_t n, float p, void* stream);

/* ---- optimizer (torch.optim.Adam semantics, utils/init_trainer.py:169-177) ------------------*/
int dcs_adam_step(float* p, const float* g, float* m, float* v, int64_t n, float lr, float beta1,
                  float beta2, float eps, float wd, int step, void* stream);

/* ---- boundary-aware label weights (dataloaders/custom_transforms_acdc.py:656-693, LabelBoundaryTransform) -----
 * labels int64 [B,H,W]; dist int32 [B,H,W] scratch -> 16.16 fixed-point 3x3 chamfer (cv2.DIST_L2, maskSize 3)
 * distance of every pixel to the nearest pixel with a different label (= the sum over classes of the per-class
 * cv2.distanceTransform values the reference computes); weight [B,H,W] = exp(-d / (2 * std_image(d))), d = 0 for
 * labels outside [0,num_classes), weight = 0 where label == ignore_id, std == 0 -> 1 (img_std [B] receives the
 * per-image standard deviation).  W <= 4096. */
int dcs_label_boundary_weights(const int64_t* labels, int32_t* dist, float* img_std, float* weight, int B, int H,
                               int W, int num_classes, int64_t ignore_id, void* stream);

/* elementwise helpers */
int dcs_axpy(float* y, const float* x, int64_t n, float a, void* stream);           /* y += a*x */
int dcs_add_rowvec_bcast(float* g, const float* v, int N, int64_t HW, int C, float scale, int accumulate,
                         void* stream);                /* g[n,p,c] (+)= scale*v[n,c]; accumulate=0 overwrites g */
int dcs_relu_bwd_rows(const float* g, const float* z, float* out, int64_t n, void* stream); /* out = g*(z>0) */

const char* dcs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* DCS_HIP_H */

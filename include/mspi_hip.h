/*
 * mspi_hip.h -- C ABI of libmspi_hip.so: the MI355X (gfx950) kernels behind MSPI's
 * saliency-inference hot path.
 *
 * The reference (oraclefina/MSPI) has no FFI boundary on this path: every op below is an
 * ATen call issued from a torch.nn.Module.forward.  Each entry point
 * therefore cites the reference call site(s) whose arithmetic it replaces; the Python
 * host (mspi_amd/) reaches them through ctypes with raw device pointers
 * (tensor.data_ptr()) and the caller's hipStream_t.  See INTEGRATION.md for the binding.
 *
 * Conventions
 *   - all tensors are fp32, device memory, owned by the caller; nothing is allocated here
 *   - activations are channels-last: a tensor [N,T,H,W,C] is a row-major matrix of
 *     M = N*T*H*W rows with a row stride `ld` (floats, multiple of 4) and C columns.
 *     ld > C lets a producer write straight into a channel slice of a concat buffer.
 *   - every function returns 0 or a negative MSPI_E* code and never throws; the message
 *     is available from mspi_last_error() (thread-local)
 *   - kernels are stateless and re-entrant; launches go to `stream` and return
 *     immediately (graph-capture safe: no sync, no allocation, no memcpy inside)
 */
#ifndef MSPI_HIP_H
#define MSPI_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MSPI_ABI_VERSION 2   /* 2: blocked plane layout (mspi_gemm_sp_fwd and friends), MspiConvDesc.w_blocked */

typedef void* mspi_stream_t; /* hipStream_t */

enum {
  MSPI_OK = 0,
  MSPI_EINVAL = -1,   /* bad descriptor (shape/stride/alignment) */
  MSPI_ELAUNCH = -2,  /* hip launch error */
  MSPI_ENODEV = -3    /* no gfx950 device / code object not loadable */
};

enum { MSPI_ACT_NONE = 0, MSPI_ACT_RELU = 1, MSPI_ACT_GELU = 2, MSPI_ACT_SIGMOID = 3, MSPI_ACT_SWISH = 4 };

/* GEMM arithmetic.  F32: v_mfma_f32_32x32x2_f32 (exact fp32).  F16X3: every fp32 operand is split into
 * hi + lo f16 halves (22 significand bits) and the product is formed as hi*hi + hi*lo + lo*hi by three
 * v_mfma_f32_32x32x16_f16 with fp32 accumulation -- fp32-level accuracy (<= 2^-21 relative per product) on
 * the 16x faster f16 matrix pipe.  Operands must satisfy |x| < 65504 and |w * w_scale| < 65504. */
enum { MSPI_PREC_F32 = 0, MSPI_PREC_F16X3 = 1 };

int mspi_version(void);
const char* mspi_last_error(void);
/* Range guard.  f16x3 operands must satisfy |x| < 65504 (beyond it the f16 hi half is inf).  The GEMM kernels (mspi_conv_fwd,
 * mspi_conv_splitk_fwd, mspi_gemm_sp_fwd, mspi_rowgemm_fwd, mspi_mlp_fwd, mspi_x3d_ab_fwd) check their pre-activation
 * results and store 1 into *word when one is inf or NaN -- whatever the cause (operand out of range, non-finite input).
 * `word` must be device-visible: a 4-byte word of pinned host memory (hipHostMalloc / torch pin_memory) lets the caller read
 * it without a device call, after the event that covers the launches; the caller clears it.  NULL (default): no report.
 * Process-global; set it before launching from several threads. */
int mspi_set_status_word(int32_t* device_visible_word);

/* number of visible HIP devices whose arch is gfx950 (0 if none). */
int mspi_device_count(void);

/* ------------------------------------------------------------------------------------
 * Dense convolution / linear as an MFMA (v_mfma_f32_32x32x2_f32) implicit GEMM.
 *   y[m, co] = act( sum_k A[m,k] * w[co,k] + bias[co] (+ res[m,co]) )
 * m runs over (n, to, ho, wo); k over (kt, kh, kw, ci) with ci fastest.
 * Replaces: nn.Conv3d / nn.Conv2d / nn.Linear with eval-mode BatchNorm folded in --
 *   SlowFast/resnet_helper.py:296-303,335-342 (X3D a / c), :427-464 (bottleneck a/b/c),
 *   :579-591 (branch1); SlowFast/stem_helper.py:262-269 (x3d conv_xy), :171-181 (basic stem);
 *   backbones/resnet.py:79-90,30-52; backbones/s3d.py:41-52,95-116;
 *   model/model_utils.py:43-46,92-94 (Linear), :324-327 (pwconv), :367-377 (smooth),
 *   :439-440 (lateral), :490-503 (readout); backbones/MViT.py:1059-1061;
 *   backbones/video_swin_transformer.py:151-153,449.
 * The input is addressed through element strides so NCDHW user tensors (clips, audio)
 * are consumed without a layout pass.  `gate` (optional, 1x1x1 stride-1 only) applies the
 * X3D squeeze-excite scale and Swish to A on the fly:
 *   A'[m,k] = swish(A[m,k] * gate[n(m), k])      (SlowFast/resnet_helper.py:66-73,76-103)
 * ------------------------------------------------------------------------------------ */
typedef struct MspiConvDesc {
  int32_t N, T, H, W, C;           /* input extent; C = channels as stored */
  int64_t sN, sT, sH, sW, sC;      /* input element strides */
  int32_t kT, kH, kW;
  int32_t strT, strH, strW;
  int32_t padT, padH, padW;
  int32_t To, Ho, Wo;              /* output extent (checked against the formula) */
  int32_t Cout;                    /* output columns written (stored width) */
  int64_t ldy;                     /* output row stride */
  int64_t ldw;                     /* weight row stride, >= kT*kH*kW*C, multiple of 4, zero padded */
  int64_t ldr;                     /* residual row stride (res != NULL) */
  int32_t act;                     /* MSPI_ACT_* applied last */
  int32_t prec;                    /* MSPI_PREC_F32: w is float [Cout][ldw];
                                      MSPI_PREC_F16X3: w is _Float16 [2][Cout][ldw] = hi/lo split of w*w_scale,
                                      ldw % 32 == 0 (see below) */
  float w_scale;                   /* F16X3: power-of-two pre-scale of the weights (undone in the epilogue) */
  int32_t tile;                    /* -1: library heuristic; else a kernel instantiation picked by the caller's
                                      autotuner: 0 128x128/4 waves, 1 128x64, 2 128x32, 3 64x64, 4 128x128/8 waves,
                                      5 256x128, 6..11 LDS-DMA kernel (128 rows, 4 waves) with 128 / 64 / all (<= 256) /
                                      96 / 192 / 32 columns per tile, 12..14 its 256-row / 8-wave form with 256 / 192 /
                                      128 columns (f16x3, 16-B gather only) */
  const void* w_blocked;           /* optional (NULL: none), F16X3 only: the same hi/lo weight planes BLOCKED as described at
                                      mspi_gemm_sp_fwd (16 output channels x 32 k = 1 KB contiguous per block, k-fastest,
                                      rows zero-padded to a multiple of 16).  The LDS-DMA kernels (tile 6..14, and the
                                      heuristic when it picks them) stage their weights from it: every piece is then 8 full
                                      cache lines instead of 16 half lines.  The other kernels read `w`. */
} MspiConvDesc;

int mspi_conv_fwd(const MspiConvDesc* d, const float* x, const float* w, const float* bias /*[Cout] or NULL*/,
                  const float* res /*NULL or [M][ldr]*/, const float* gate /*NULL or [N][C]*/,
                  float* y, mspi_stream_t stream);

/* Split-K form of mspi_conv_fwd for problems with few output tiles and a long contraction (M*Cout small, K large:
 * the SA / smoothing convs on 14x14 maps, the audio ResNet's last stages, SlowFast s5): ksplit workgroups share each
 * 64x64 output tile, each owns a contiguous range of K steps and writes its partial sums to the workspace
 * (mspi_conv_splitk_ws_bytes(d, ksplit) bytes); a second launch adds the slices in order and applies bias, residual and
 * activation -- bitwise reproducible, no atomics.  No gate; Cout % 4 == 0. */
size_t mspi_conv_splitk_ws_bytes(const MspiConvDesc* d, int32_t ksplit);
int mspi_conv_splitk_fwd(const MspiConvDesc* d, const float* x, const float* w, const float* bias, const float* res,
                         float* y, void* workspace, int32_t ksplit, mspi_stream_t stream);

/* Which kernel instantiation the calling thread's last mspi_conv_fwd launched:
 * (BM << 16) | (BN << 4) | (8 if 8 waves) | (4 if LDS-DMA staging) | (prec << 1) | (1 if scalar gather, 0 if
 * 16-B vector gather).  For profiling: it names the template instantiation rocprofv3 reports. */
int mspi_conv_last_config(void);

/* ------------------------------------------------------------------------------------
 * Depthwise convolution, channels-last, bias (= folded BN) + activation fused; optional
 * per-(n,c) PARTIAL sums of the pre-activation output for squeeze-excite: pool is
 * [N][mspi_dwconv_pool_rows(d)][C], one row per workgroup, written (not accumulated) in a
 * fixed order -- no atomics, bitwise reproducible; mspi_se_gate reduces the rows.
 * Replaces: X3D `b` 3x3x3 (SlowFast/resnet_helper.py:310-319) + b_bn (+ Swish :76-103),
 *   X3D stem (5,1,1) (SlowFast/stem_helper.py:270-283), ConvNextBlock.dwconv_t/dwconv_s
 *   (model/model_utils.py:321-322), MViT pool_q/k/v (backbones/MViT.py:1093-1133),
 *   timm ConvNeXt conv_dw 7x7.
 * w is [kT*kH*kW][C] (tap-major), x rows have stride ldx, y rows ldy.
 * ------------------------------------------------------------------------------------ */
typedef struct MspiDwConvDesc {
  int32_t N, T, H, W, C;
  int64_t ldx, ldy;
  int32_t kT, kH, kW;
  int32_t strT, strH, strW;
  int32_t padT, padH, padW;
  int32_t To, Ho, Wo;
  int32_t act;
} MspiDwConvDesc;

int mspi_dwconv_fwd(const MspiDwConvDesc* d, const float* x, const float* w, const float* bias,
                    float* y, float* pool /*NULL or [N][rows][C]*/, mspi_stream_t stream);
/* partial-sum rows per sample that mspi_dwconv_fwd writes for this descriptor (-1: pooling unsupported) */
int mspi_dwconv_pool_rows(const MspiDwConvDesc* d);

/* Squeeze-excite gate: gate[n,c] = sigmoid(fc2(relu(fc1(inv_count * sum_r pool[n,r,:]))))
 * (SlowFast/resnet_helper.py:27-73).  w1 [F][C], b1 [F], w2 [C][F], b2 [C]. */
int mspi_se_gate(const float* pool, int32_t rows, float inv_count, const float* w1, const float* b1,
                 const float* w2, const float* b2, float* gate, int32_t N, int32_t C, int32_t F,
                 mspi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * LayerNorm over the C columns of each row, one wavefront per row:
 *   y[n,r,:] = act( (x[n,r,:]-mean)/sqrt(var+eps) * gamma + beta ) + table[r, :]
 * Rows are addressed as base + n*sN + r*ld for n < N, r < R on both sides, so a producer can
 * write token slabs of a [N, R_total, C] sequence buffer (the torch.cat at
 * model/model_utils.py:277 disappears).  table (optional) has R rows.
 * Replaces nn.LayerNorm at model/model_utils.py:231-233 (+ sinusoid table add :273-274),
 *   :139,145 (Block), :296 (LayerNorm3d), :404-435 (projector LN+ReLU);
 *   backbones/MViT.py:1714; backbones/video_swin_transformer.py:306; timm LayerNorm2d.
 * ------------------------------------------------------------------------------------ */
int mspi_layernorm_fwd(const float* x, int64_t ldx, int64_t sNx, float* y, int64_t ldy, int64_t sNy,
                       const float* gamma, const float* beta, float eps, int32_t N, int32_t R, int32_t C,
                       int32_t act, const float* table /*NULL or [R][C]*/, mspi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Fused multi-head attention (flash style, MFMA, online softmax, fp32 in / fp32 accumulate):
 *   o[b,h,i,:] = softmax_j( scale * q[b,h,i,:] . k[b,h,j,:] + biasT[h,j,i] + maskT[b % nmask,j,i] ) v[b,h,j,:] (+ res)
 * q/k/v/o (and res, with o's strides) are addressed as base + b*sB + h*sH + token*sT + d (d contiguous).
 * D = head dim of q/k, Dv = head dim of v/o; (D,Dv) in {(32,32),(64,64),(96,96),(128,128),(128,96),(160,96)}.
 * biasT / maskT (optional) are stored key-major ([.][Nk][Nq]).
 * tok_idx (optional, [nwin][N] int32): windowed sequences -- sequence b is window b % nwin of sample b / nwin
 * (strides sB then address the SAMPLE) and its token t lives at row tok_idx[b % nwin][t] of that sample, for
 * q, k, v, res and o alike: Swin's cyclic shift, window partition, window reverse and un-shift
 * (backbones/video_swin_transformer.py:61-87,245-268) become index arithmetic inside the kernel.
 * Replaces model/model_utils.py:102-106 (SyncBlock), backbones/MViT.py:1261-1301 (pooled attention, residual
 * pooling as `res`), backbones/video_swin_transformer.py:169-187 (window attention, bias table + shift mask).
 * ------------------------------------------------------------------------------------ */
typedef struct MspiAttnDesc {
  int32_t B, Hh, Nq, Nk, D, Dv, nmask, nwin;
  int64_t q_sB, q_sH, q_sT;
  int64_t k_sB, k_sH, k_sT;
  int64_t v_sB, v_sH, v_sT;
  int64_t o_sB, o_sH, o_sT;
  float scale;
  int32_t prec;   /* MSPI_PREC_F32: fp32 MFMA; MSPI_PREC_F16X3: split products on the f16 pipe (fp32-accurate) */
} MspiAttnDesc;

int mspi_attn_fwd(const MspiAttnDesc* d, const float* q, const float* k, const float* v, const float* res,
                  const float* biasT, const float* maskT, const int32_t* tok_idx, float* o, mspi_stream_t stream);

/* The same attention (f16x3 only) with K and V split ONCE per (sequence, head) into f16 hi/lo planes in a caller-owned
 * workspace of mspi_attn_ws_bytes(d) bytes, instead of by every query tile of that head on its own copy: two launches (plane
 * kernel, attention kernel), results bit-identical to mspi_attn_fwd.  mspi_attn_ws_bytes returns 0 for other precisions.
 * Without bias, mask and token index the attention kernel is the software-pipelined form (csrc/attn.hip, attn_pipe_kernel:
 * the planes are then per-tile LDS images staged by LDS-DMA); same products in the same order, same results.
 * Few-query shapes (fewer than 256 query tiles over all heads, at least 24 key tiles; no bias, mask or token index) are
 * additionally split along the keys: up to 8 workgroups per query tile each walk a slice of the key tiles and a third
 * launch merges their (O, running max, running sum) in fixed order -- deterministic, equal to the one-pass result up to
 * fp32 rounding of the merge; the workspace size accounts for the partial results.  MSPI_ATTN_KSPLIT=0 switches it off. */
size_t mspi_attn_ws_bytes(const MspiAttnDesc* d);
int mspi_attn_fwd_ws(const MspiAttnDesc* d, const float* q, const float* k, const float* v, const float* res,
                     const float* biasT, const float* maskT, const int32_t* tok_idx, float* o, void* workspace,
                     mspi_stream_t stream);

/* MViTv2 decomposed relative positions folded into the attention contraction (backbones/MViT.py:905-997):
 *   qa[b,h,i,:] = [ scale*q_i | q_i.Rh[hq(i),0..kH) | q_i.Rw[wq(i),0..kW) | q_i.Rt[tq(i),0..kT) | 0 ]   (DA columns)
 *   ka[b,h,j,:] = [ k_j | onehot_kH(hk(j)) | onehot_kW(wk(j)) | onehot_kT(tk(j)) | 0 ]
 * so that qa.ka^T = scale*q.k + rel_h + rel_w + rel_t exactly; feed qa/ka to mspi_attn_fwd with D = DA, scale 1.
 * q rows are [B*Nq][ldq] with head h at column h*Dh (likewise k); Rh/Rw/Rt are the gathered tables
 * [qH][kH][Dh], [qW][kW][Dh], [qT][kT][Dh]. */
typedef struct MspiMvitAugDesc {
  int32_t B, heads, Dh, DA;
  int32_t qT, qH, qW, kT, kH, kW;
  int64_t ldq, ldk;
  float scale;
} MspiMvitAugDesc;

int mspi_mvit_qk_augment(const MspiMvitAugDesc* d, const float* q, const float* k, const float* Rh, const float* Rw,
                         const float* Rt, float* qa, float* ka, mspi_stream_t stream);

/* Same result with the dot products done by a GEMM: P[(b*Nq + tok)*heads + head][ldp] = q_row . T^T, T = the rows of the
 * three (length-matched) relative-position tables stacked; idx_h [qH][kH], idx_w [qW][kW], idx_t [qT][kT] give the column
 * of P that holds q . R*[position, j] (the relative distance of backbones/MViT.py:905-990 plus the table's offset).  The
 * caller computes P with mspi_conv_fwd / mspi_rowgemm_fwd on the q rows; this call only copies and gathers. */
int mspi_mvit_qk_augment_p(const MspiMvitAugDesc* d, const float* q, const float* k, const float* P, int64_t ldp,
                           const int32_t* idx_h, const int32_t* idx_w, const int32_t* idx_t, float* qa, float* ka,
                           mspi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * X3D block, first half, fused (csrc/x3d_block.hip):
 *   u = act( b_bn( dw3x3x3( relu( a_bn( a(x) ) ) ) ) ),  act = SWISH (blocks without squeeze-excite) or NONE (+ pool)
 * Replaces X3DTransform's a, a_bn, a_relu, b, b_bn (SlowFast/resnet_helper.py:296-319) for the stride-1 blocks; the
 * 2.25x-wide `a` output stays in LDS.  x: [N,T,H,W] rows of ldx floats (Cin stored channels, Cin % 8 == 0); u: rows of
 * ldu floats (Cmid stored channels).  wa_packed: mspi_x3d_ab_packed_bytes(Cin, Cmid) bytes, f16 hi/lo of a's weights
 * (BN folded) times wa_scale in MFMA fragment order [chunk of 32 outputs][k32 step][16-row half][hi,lo][lane][8]: element
 * e of lane l = W[chunk*32 + half*16 + (l & 15)][32*step + 8*(l >> 4) + e], zero padded.  wb: fp32 [27][Cmid] taps
 * (kt,kh,kw major) with b_bn folded, bias_a / bias_b fp32.  pool (optional): [N][mspi_x3d_ab_pool_rows(d)][Cmid] partial
 * sums of the PRE-activation output for mspi_se_gate (one row per workgroup, no atomics).
 * H % 7 == 0 and (W % 14 == 0 or W == 7); Cin <= 96 or Cin in (160, 192]. */
typedef struct MspiX3dAbDesc {
  int32_t N, T, H, W;
  int32_t Cin, Cmid;               /* stored channel counts */
  int64_t ldx, ldu;
  int32_t act;                     /* MSPI_ACT_NONE or MSPI_ACT_SWISH, applied to u (not to the pooled sums) */
  float wa_scale;                  /* power-of-two pre-scale of a's weights */
} MspiX3dAbDesc;

int mspi_x3d_ab_supported(const MspiX3dAbDesc* d);
int mspi_x3d_ab_pool_rows(const MspiX3dAbDesc* d);
size_t mspi_x3d_ab_packed_bytes(int32_t Cin, int32_t Cmid);
int mspi_x3d_ab_fwd(const MspiX3dAbDesc* d, const void* x, const void* wa_packed, const void* bias_a, const void* wb,
                    const void* bias_b, void* u, void* pool /*or NULL*/, mspi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Max pooling, channels-last, -inf padding.
 * Replaces nn.MaxPool3d / MaxPool2d at model/model_utils.py:189,206;
 *   backbones/resnet.py:82; SlowFast/stem_helper.py:195-197; backbones/MViT.py:1403-1409.
 * ------------------------------------------------------------------------------------ */
int mspi_maxpool_fwd(const MspiDwConvDesc* d, const float* x, float* y, mspi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Bilinear spatial up-sampling by an integer factor (align_corners=False; T untouched,
 * which is what trilinear with scale (1,k,k) computes), optionally accumulating:
 *   dst[n,t,ho,wo,:] = act( (dst[n,t,ho,wo,:] +) bilinear(src[n,t,:,:,:]) )
 * Replaces nn.Upsample at model/model_utils.py:158,208,486-488,498 and the adds at :566-570.
 * ------------------------------------------------------------------------------------ */
int mspi_upsample_fwd(const float* src, int64_t lds, float* dst, int64_t ldd, int32_t NT, int32_t H,
                      int32_t W, int32_t C, int32_t factor, int32_t accumulate, int32_t act,
                      mspi_stream_t stream);

/* 2x2 spatial space-to-depth of Swin's PatchMerging (backbones/video_swin_transformer.py:311-326):
 * y[n,t,h,w, q*C + c] = x[n,t,2h+dh(q),2w+dw(q),c] with (dh,dw)(q) = (0,0),(1,0),(0,1),(1,1).  H, W even. */
int mspi_space_to_depth(const float* x, int64_t ldx, float* y, int64_t ldy, int32_t NT, int32_t H, int32_t W,
                        int32_t C, mspi_stream_t stream);

/* SA gating x*m + x (model/model_utils.py:167-170): x[m,:] *= (1 + mask[m]), in place. */
int mspi_rowgate(float* x, int64_t ldx, const float* mask, int64_t M, int32_t C, mspi_stream_t stream);

/* out[n,:] -= logsumexp(out[n,:]) over L elements per sample (model/model_utils.py:572). */
int mspi_logsumexp_sub(float* x, int32_t N, int32_t L, mspi_stream_t stream);

/* out[n,c] = mean over R rows of x[n,r,c] (nn.AdaptiveAvgPool, model/model_utils.py:402-403,543-544). */
int mspi_mean_rows(const float* x, int64_t ldx, int64_t rows_per_sample_stride, float* out, int32_t N,
                   int32_t R, int32_t C, mspi_stream_t stream);

/* The same mean for long samples (MorphFC re-weighting, backbones/MorphMLP.py:62,104: 25088 rows per sample): two
 * deterministic stages through a caller-owned workspace of N * mspi_mean_rows_slices(R) * C floats.
 * mspi_mean_rows_slices returns 0 when the one-stage mspi_mean_rows is the right call (R < 1024). */
int mspi_mean_rows_slices(int32_t R);
int mspi_mean_rows_ws(const float* x, int64_t ldx, int64_t rows_per_sample_stride, float* out, float* ws, int32_t N,
                      int32_t R, int32_t C, mspi_stream_t stream);

/* out[0] (+)= scale * mean_n( -cos(p[n,:], z[n,:]) )   (D(), model/model_utils.py:285-290). */
int mspi_neg_cosine(const float* p, const float* z, float* out, int32_t N, int32_t C, float scale,
                    int32_t accumulate, mspi_stream_t stream);

/* Saliency-map post-processing (inference.py:66-69,85-89; OpenCV upstream -- parity unpinned):
 * out[n] = uint8( round( 255 * minmax( resize_bilinear( exp( GaussianBlur11x11(logmap[n]) ), Ho x Wo ) ) ) ).
 * workspace: mspi_postprocess_workspace(...) bytes of device memory. */
size_t mspi_postprocess_workspace(int32_t N, int32_t H, int32_t W, int32_t Ho, int32_t Wo);
int mspi_postprocess_u8(const float* logmap, unsigned char* out, void* workspace, int32_t N, int32_t H, int32_t W,
                        int32_t Ho, int32_t Wo, mspi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Fused channel MLP on rows:  y = res + W2 . act( W1 . LN(x) + b1 ) + b2,  the 4C-wide hidden
 * activation stays on the CU (csrc/mlp_fused.hip).  f16x3 split products, fp32 accumulate.
 * Replaces: timm ConvNeXt block norm -> mlp.fc1 -> GELU -> mlp.fc2 -> gamma -> + shortcut
 *   (via model/model_utils.py:361,380), and the LN -> Mlp -> residual tail of
 *   SwinTransformerBlock3D.forward_part2 (backbones/video_swin_transformer.py:262-263) and
 *   MultiScaleBlock (backbones/MViT.py:1420-1432) when dim_out == dim.
 * C in {96, 192}; hidden a multiple of 32, <= 1024; x, res, y row-major with strides ldx/ldr/ldy.
 * w_packed: mspi_mlp_packed_bytes(C, hidden) bytes of f16, per hidden chunk j of 32 units
 *   W1 part [ks < C/16][hi,lo][lane < 64][e < 8] = W1s[j*32 + lane%32][16 ks + 8 (lane/32) + e]
 *   W2 part [s < 2][ct < C/32][hi,lo][lane][e]   = W2s[ct*32 + lane%32][j*32 + (2s + e/4)*8 + 4 (lane/32) + e%4]
 * with W1s = w1_scale * fc1.weight [hidden, C], W2s = w2_scale * (out_scale (.) fc2.weight) [C, hidden],
 * hi = f16(Ws), lo = f16(Ws - hi)  (engine.pack_mlp builds it). */
typedef struct {
  int64_t M;                 /* rows */
  int32_t C, hidden;
  int64_t ldx, ldr, ldy;     /* row strides in floats */
  int32_t ln;                /* 1: LayerNorm over C (gamma, beta, eps) applied to x first */
  int32_t act;               /* MSPI_ACT_* between the two layers */
  float eps;
  float w1_scale, w2_scale;  /* powers of two the packed weights were multiplied by */
} MspiMlpDesc;
size_t mspi_mlp_packed_bytes(int32_t C, int32_t hidden);
int mspi_mlp_fwd(const MspiMlpDesc* d, const void* x, const void* gamma, const void* beta, const void* w_packed,
                 const void* b1, const void* b2, const void* res, void* y, mspi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * Row-stationary thin GEMM (1x1x1 conv / Linear with K <= 224 and few output columns):
 *   y[M, N] = act( x'[M, K] . W^T + bias (+ res) ),   x' = x, or swish(x * gate[row / rows_per_sample]) when gate != NULL.
 * Same contract as mspi_conv_fwd on a dense 1x1x1 stride-1 problem; a different kernel (csrc/mlp_fused.hip) and a
 * different weight packing.  Replaces: X3DTransform.a / .c (SlowFast/resnet_helper.py:296-351, with the SE scale :333
 * and Swish :339 as the gate prologue), ResBlock.branch1 (:540-556), X3D conv5 pieces (backbones/X3D.py).
 * K, N: storage columns of x / y (multiples of 4, pad columns zero).  mspi_rowgemm_supported(K, N) says whether the
 * shape is covered (K <= 224, N <= 1024).
 * w_packed: mspi_rowgemm_packed_bytes(K, N) bytes of f16; with KSB = 2, 4, 8 or 14 k-steps of 16 (the smallest
 * covering K) and Ws = w_scale * W [N, K] zero-padded:  [chunk j < ceil(N/32)][ks < KSB][hi,lo][lane < 64][e < 8]
 *   = Ws[j*32 + lane%32][16 ks + 8 (lane/32) + e]   (engine.pack_rowgemm builds it). */
typedef struct {
  int64_t M;
  int32_t K, N;
  int64_t ldx, ldr, ldy, ldg;   /* row strides in floats (ldg: gate rows) */
  int32_t act;
  int32_t rows_per_sample;      /* gate row index = row / rows_per_sample */
  float w_scale;
} MspiRowGemmDesc;
size_t mspi_rowgemm_packed_bytes(int32_t K, int32_t N);
int mspi_rowgemm_supported(int32_t K, int32_t N);
int mspi_rowgemm_fwd(const MspiRowGemmDesc* d, const void* x, const void* w_packed, const void* bias, const void* res,
                     const void* gate, void* y, mspi_stream_t stream);

/* ------------------------------------------------------------------------------------
 * X3D block seam, one launch for the end of block i and the start of block i+1 of a stage (csrc/mlp_fused.hip):
 *   y[M, Cx] = relu( u'[M, D] . Wc^T + bc + res ),   u' = u, or swish(u * gate[row / rows_per_sample]) when gate != NULL
 *   t[M, D]  = relu( y . Wa^T + ba )
 * Replaces: X3DTransform.c + c_bn, the residual add and ReLU of ResBlock.forward (SlowFast/resnet_helper.py:339-351,
 * :607-616), followed by the next block's X3DTransform.a + a_bn + a_relu (:296-307); the SE scale (:333) and Swish
 * (:339) of block i are the gate prologue, as in mspi_rowgemm_fwd.  Results are those of the two mspi_rowgemm_fwd
 * calls it stands for, up to fp32 summation order.
 * D: stored columns of u and t (the stage's inner width, <= 224), Cx: stored columns of res and y (<= 256); multiples of 4.
 * w_packed: mspi_x3d_ca_packed_bytes(D, Cx) bytes = mspi_mlp_fwd's packing with C = D padded to 128 or 224,
 * hidden = Cx padded to 32, W1s = wc_scale * Wc [Cx, D], W2s = wa_scale * Wa [D, Cx], zero padding (engine.pack_x3d_ca). */
typedef struct {
  int64_t M;
  int32_t D, Cx;
  int64_t ldu, ldr, ldy, ldt, ldg;   /* row strides in floats */
  int32_t rows_per_sample;
  float wc_scale, wa_scale;
} MspiX3dCaDesc;
size_t mspi_x3d_ca_packed_bytes(int32_t D, int32_t Cx);
int mspi_x3d_ca_supported(int32_t D, int32_t Cx);
int mspi_x3d_ca_fwd(const MspiX3dCaDesc* d, const void* u, const void* gate, const void* w_packed, const void* bc,
                    const void* ba, const void* res, void* y, void* t, mspi_stream_t stream);

/* Saliency metrics (utils/compute_saliency_metrics.py:9-108; the terms of utils/loss.py:26-49): per sample n,
 * out[n] = { KL(gt || pred), CC(pred, gt), SIM(pred, gt), NSS(pred, fix) } over the L = H*W values of each map.
 * pred is the predicted map (pred_is_log: the model's log-probability map, exponentiated on the fly), gt the
 * ground-truth density, fix the binary fixation map (NULL: NSS is written as 0).  Batch means are the caller's. */
int mspi_saliency_metrics(const float* pred, const float* gt, const float* fix, float* out /*[N][4]*/, int32_t N, int32_t L,
                          int32_t pred_is_log, mspi_stream_t stream);

/* MorphMLP token regrouping (backbones/MorphMLP.py:49-58,87-100,134-137: the reshape/permute/reshape chains around
 * mlp_h / mlp_w / mlp_t) as ONE strided gather: y is dense with extents dims[0..5] (dims[5] innermost),
 * y[i0..i5] = x[sum_k i_k * strides[k]]; strides[5] must be 1, src_elems bounds the reads. */
typedef struct MspiPermuteDesc {
  int32_t dims[6];
  int64_t strides[6];
  int64_t src_elems;
} MspiPermuteDesc;
int mspi_permute_fwd(const MspiPermuteDesc* d, const float* x, float* y, mspi_stream_t stream);

/* MorphFC re-weighting (backbones/MorphMLP.py:64-67,104-107): y[n,r,c] = sum_j softmax_j(logit[n, c*J + j]) * src_j[n,r,c]
 * over dense [N, rows_per_sample, C] operands; J = 3 (a, b, c) or 2 (a, b; c may be NULL). */
int mspi_gated_sum_fwd(const float* a, const float* b, const float* c, const float* logit, float* y, int32_t N,
                       int64_t rows_per_sample, int32_t C, int32_t J, mspi_stream_t stream);

/* Log-spectrogram windows of the clip loop (inference.py:24-63: torchaudio Spectrogram(n_fft=512, hop_length=160) on
 * audio[start:end] (optionally time-reversed), log(p + 1e-6), per-column standardisation over the 257 bins with the
 * unbiased std, crop / pad with 0.02 to Wa columns).  wave: 16 kHz mono samples on the device; seg [B][3] =
 * (start, length, reversed) on the device, seg_host the same table on the host (bounds are validated before the launch);
 * window: 512 Hann coefficients on the device; out [B][257][Wa]. */
int mspi_logspec_fwd(const float* wave, int64_t n_wave, const int32_t* seg, const int32_t* seg_host, int32_t B,
                     const float* window, float* out, int32_t Wa, mspi_stream_t stream);

/* Frame pre-processing (inference.py:154-165: torchvision Resize on a PIL image = PIL's antialiased bilinear resampling,
 * ToTensor, Normalize).  rgb: uint8 [Hin][Win][3] on the device; tmp: Hin*Wout*3 bytes of device scratch; out: fp32
 * [3][Hout][Wout] planes `out_plane_stride` floats apart.  hb/vb: [n][2] = (first input index, taps) per output
 * column / row, hk/vk: [n][hks|vks] 22-bit fixed-point taps -- PIL's precompute_coeffs + normalize_coeffs_8bpc tables,
 * built by the caller (mspi_amd/preproc.py) and resident on the device; mean3/std3: host floats. */
int mspi_resize_norm_fwd(const unsigned char* rgb, int32_t Hin, int32_t Win, unsigned char* tmp, float* out,
                         int64_t out_plane_stride, int32_t Hout, int32_t Wout, const int32_t* hb, const int32_t* hk,
                         int32_t hks, const int32_t* vb, const int32_t* vk, int32_t vks, const float* mean3_host,
                         const float* std3_host, mspi_stream_t stream);

/* Pre-split activations.  The f16x3 GEMM computes x*w from f16 hi/lo halves of both operands; the weights are split at pack
 * time, and a producer may hand the activations over ALREADY split: two f16 planes (hi plane, lo plane `plane` elements
 * later; hi = f16(x), lo = f16(x - hi) -- the same split the kernels otherwise do in registers, so results are
 * bit-identical).  Each plane is BLOCKED: K % 32 == 0, the row count is padded to a multiple of 16, and element (m, k) sits at
 *     ((m / 16) * (K / 32) + k / 32) * 512 + (m % 16) * 32 + k % 32        (halves)
 * i.e. a 16-row x 32-column block is 1 KB contiguous and the blocks of a row group follow each other along k: the k32 stage
 * of a 16-row group is one contiguous 1-KB LDS-DMA piece of 8 full cache lines (row-major planes hand the loader 16 half
 * lines per piece: 30 instead of 43 B/clk/CU of fill, tools/dma_issue_probe.hip).  `ld` arguments of planes must equal K;
 * `plane` >= roundup16(M) * K; the pad rows are read (never stored): keep them finite.
 * mspi_gemm_sp_fwd is the dense (1x1x1 / nn.Linear) GEMM on such planes: both operands go HBM -> LDS -> MFMA with no
 * conversion work in the loop.  d: as for mspi_conv_fwd with C % 32 == 0, ldw == C, prec f16x3; `w`: the f16 hi/lo weight
 * planes of mspi_conv_fwd BLOCKED the same way (rows = output channels, zero-padded to a multiple of 16; lo plane
 * roundup16(Cout) * K halves after the hi plane; engine.sp_weights builds it);
 * d->tile: -1 heuristic, 6/7/9/10/11 = 128 x {128,64,96,192,256}, 12/13/14 = 256 x {256,192,128}.  The result goes to y
 * (fp32 rows, ldy) or, when y_planes != NULL, to blocked output planes (ldys == Cout, Cout % 32 == 0) for the next GEMM.
 * mspi_split_planes_fwd converts fp32 rows. */
int mspi_layernorm_sp_fwd(const float* x, int64_t ldx, int64_t sample_stride_x, void* planes, int64_t ldo, int64_t plane,
                          const float* gamma, const float* beta, float eps, int32_t N, int32_t R, int32_t C, int32_t act,
                          mspi_stream_t stream);   /* mspi_layernorm_fwd writing pre-split planes (rows dense, n*R + r) */
int mspi_split_planes_fwd(const float* x, int64_t ldx, int64_t M, int32_t K, void* planes, int64_t ldo, int64_t plane,
                          mspi_stream_t stream);
/* planes -> fp32 rows (hi + lo): for a consumer that was moved off the f16x3 path after its producer had emitted planes */
int mspi_join_planes_fwd(const void* planes, int64_t ldi, int64_t plane, int64_t M, int32_t K, float* y, int64_t ldy,
                         mspi_stream_t stream);
int mspi_gemm_sp_fwd(const MspiConvDesc* d, const void* x_planes, int64_t ldx, int64_t xplane, const float* w,
                     const float* bias, const float* res, float* y, void* y_planes, int64_t ldys, int64_t yplane,
                     mspi_stream_t stream);

/* y = a + b over n floats (plain residual add where no producer can fuse it). */
int mspi_add(const float* a, const float* b, float* y, int64_t n, mspi_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* MSPI_HIP_H */

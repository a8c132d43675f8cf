/* skg.h - C ABI of libskg.so: MI355X (gfx950) kernels for the sketch-guided diffusion
 * sampler hot path (Mikubill/sketch2img).
 *
 * The reference is 100 % Python and has no FFI of its own; every "kernel" on its hot path is
 * reached through third-party Python packages (diffusers -> torch/cuDNN/cuBLAS, xformers).  Each
 * entry point below therefore cites the reference CALL SITE (file:line under the reference
 * repo root) whose third-party kernels it replaces.  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *   - all pointers are DEVICE pointers unless noted; the caller owns every buffer (the library
 *     never allocates or frees), `stream` is a hipStream_t passed as void* (0 = null stream);
 *   - activations are fp16, token-major / NHWC: a tensor of `rows` images of H x W pixels and C
 *     channels is the row-major matrix [rows*H*W][C] with an explicit leading dimension (in
 *     elements) where a slice of a wider buffer may be passed;
 *   - weights are fp16 [N][K] row-major ("out x in", K contiguous), see each function;
 *   - return value: 0 = launched, negative = SKG_E_* (nothing launched).  No exceptions.
 */
#ifndef SKG_H_
#define SKG_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SKG_OK 0
#define SKG_E_BADARG (-1)    /* shape/alignment precondition violated */
#define SKG_E_UNSUPPORTED (-2)
#define SKG_E_LAUNCH (-3)    /* hipGetLastError() != hipSuccess after the launch */

#define SKG_ABI_VERSION 5
int skg_abi_version(void);
/* Human-readable text of the last SKG_E_LAUNCH on this thread ("" if none). */
const char* skg_last_error(void);

/* ---- epilogue flags shared by skg_gemm_f16 / skg_conv3x3_f16 ---------------------------------- */
#define SKG_EPI_RELU 1u      /* max(.,0) applied last */
#define SKG_EPI_OUT_F32 2u   /* C is float* instead of fp16 */
#define SKG_EPI_GEGLU 4u     /* skg_gemm_f16 only: B / bias rows are the interleaved FF1 pack (groups of four output
                              * columns [a a g g]); C is [M][N/2] = a * gelu(g).  No residual, fp16 out, K % 64 == 0
                              * (else SKG_E_UNSUPPORTED).  Fuses diffusers GEGLU into ff.net.0.proj. */

/* C[m][n] = epi( alpha * (sum_k A[m][k] * B[n][k] + bias[n]) + residual[m][n] )
 * A fp16 [M][K] (lda), B fp16 [N][K] (ldb), C fp16|fp32 [M][N] (ldc), bias fp16 [N] or NULL,
 * residual fp16 [M][N] (ldr) or NULL.  Requires K % 32 == 0, N % 8 == 0, lda/ldb % 8 == 0,
 * ldc/ldr % 4 == 0, 16-byte aligned A/B, 8-byte aligned C/residual/bias.
 * Replaces: every nn.Linear / 1x1 Conv2d the UNet evaluates at modules/pipeline.py:96
 * (diffusers proj_in/proj_out, to_q/k/v, to_out, ff.net, conv_shortcut -> cuBLAS/cuDNN), their
 * backward-to-input GEMMs triggered at modules/pipeline.py:159, the LGP's Linear layers at
 * modules/latent_predictor.py:45 and Conv1d(k=1) at modules/clip_guided_attn.py:124. */
int skg_gemm_f16(const void* A, int lda, const void* B, int ldb, void* C, int ldc,
                 int M, int N, int K, const void* bias, const void* residual, int ldr,
                 float alpha, unsigned flags, void* stream);

/* Optional split-K workspace for skg_gemm_f16 / skg_conv3x3_f16 launches on `stream` of the CURRENT device: a caller-owned
 * device buffer (>= 1 MiB, 16-byte aligned, fp32 partial slabs) that must stay valid until replaced.  With it, launches that
 * would put fewer than one workgroup on every CU (the 8x8 / 16x16-resolution layers) split their K loop over several
 * workgroups per tile and a second tiny kernel sums the slabs and applies the epilogue.  One slab per stream: launches on
 * one stream are ordered and share theirs, launches on different streams never do (the registry is mutex-protected and
 * the pointer reaches the kernel as an argument).  ws = NULL removes the stream's entry. */
int skg_set_workspace(void* ws, size_t bytes, void* stream);

/* Which kernel instantiation skg_gemm_f16 (mode 0, Cin ignored) / skg_conv3x3_f16 (mode = 1 + SKG_CONV_*)
 * run for this shape: 2000 + BN for the LDS-DMA kernel (gemm2_kernel<BN,MODE>), 1000 + BN for the generic one;
 * lets a profiler attribute a launch to the kernel instantiation that ran (bench.py roofline). */
int skg_gemm_variant(int M, int N, int K, int Cin, int mode);

/* 3x3 convolution, padding 1, as implicit GEMM over NHWC fp16.
 *   mode SKG_CONV_S1      stride 1                         out (OH,OW) = in (IH,IW)
 *   mode SKG_CONV_S2      stride 2                         out = in / 2
 *   mode SKG_CONV_UP2     nearest 2x upsample then stride 1  out = 2 * in
 *   mode SKG_CONV_S2T     transpose of S2 (its dgrad)      out = 2 * in
 *   mode SKG_CONV_S2A     stride 2, padding (0,1,0,1): bottom/right only (the VAE encoder's Downsample2D)  out = in / 2
 * X fp16 [rows*IH*IW][Cin] (ldx), Wp fp16 [Cout][3][3][Cin] (tap-major, Cin contiguous),
 * Y [rows*OH*OW][Cout] (ldy).  IH, IW are the INPUT sizes.  Requires Cin % 32 == 0, Cout % 8 == 0.
 * bias / residual / alpha / flags as skg_gemm_f16.  For a dgrad pass Wp is the flipped,
 * in/out-swapped pack (see sketch2img_amd/weights.py).
 * Replaces: ResnetBlock2D conv1/conv2, Downsample2D, Upsample2D, conv_in/conv_out (cuDNN) at
 * modules/pipeline.py:96 and their dgrad at modules/pipeline.py:159. */
#define SKG_CONV_S1 0
#define SKG_CONV_S2 1
#define SKG_CONV_UP2 2
#define SKG_CONV_S2T 3
#define SKG_CONV_S2A 4
int skg_conv3x3_f16(const void* X, int ldx, const void* Wp, void* Y, int ldy,
                    int rows, int IH, int IW, int Cin, int Cout, int mode,
                    const void* bias, const void* residual, int ldr, float alpha,
                    unsigned flags, void* stream);

/* ---- GroupNorm (+SiLU), NHWC fp16 -------------------------------------------------------------
 * stats: per (row, group) mean and rstd (float2 [rows][groups]) over HW x (C/groups) values.
 *   `partial` is scratch of skg_groupnorm_scratch_floats(rows, groups) floats.
 * apply: y = act((x-mean)*rstd*gamma+beta), act = SiLU if silu != 0.
 * bwd:   dx = GN'(SiLU'(dy)) + residual (residual may be NULL); needs x and the saved stats.
 * Replaces: torch group_norm + silu (ResnetBlock2D.norm1/2 + nonlinearity, Transformer2DModel.norm,
 * conv_norm_out) at modules/pipeline.py:96; backward at :159. */
size_t skg_groupnorm_scratch_floats(int rows, int groups);
int skg_groupnorm_stats(const void* X, int ldx, int rows, int HW, int C, int groups, float eps,
                        float* stats, float* partial, void* stream);
/* stats + apply in two launches (the apply kernel folds the chunk partials itself and publishes `stats` for a later
 * skg_groupnorm_bwd): what the UNet / VAE forward use. */
int skg_groupnorm_fwd(const void* X, int ldx, void* Y, int ldy, int rows, int HW, int C, int groups, float eps,
                      const void* gamma, const void* beta, int silu, float* stats, float* partial, void* stream);
/* GroupNorm whose statistics pass is done by the PRODUCER of X.  skg_gemm_f16_gn / skg_conv3x3_f16_gn are
 * skg_gemm_f16 / skg_conv3x3_f16 that additionally leave, per (sample b, 128-row chunk c, group g), sum(y) and sum(y^2)
 * of the fp16 outputs in gn_partial[((b * (HW / 128) + c) * groups + g) * 2 + {0, 1}] (HW = rows of C per sample, a
 * multiple of 128; fp16 output, no fused GEGLU).  The sums come out of the kernel's own epilogue when the tile that runs
 * can produce them (256 x 320 and 128 x 160 tiles, whole tiles, groups inside a tile - every 64 x 64 / 32 x 32-level
 * layer of the UNet), otherwise from the stand-alone statistics pass: the result is the same either way, in a fixed
 * summation order.  skg_groupnorm_from_partial folds `nch` chunks per (sample, group), publishes (mean, rstd) to `stats`
 * and applies - X is read once instead of twice.  Same replaced reference calls as skg_groupnorm_fwd. */
int skg_gemm_f16_gn(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                    const void* bias, const void* residual, int ldr, float alpha, unsigned flags,
                    float* gn_partial, int HW, int groups, void* stream);
int skg_conv3x3_f16_gn(const void* X, int ldx, const void* Wp, void* Y, int ldy, int rows, int IH, int IW,
                       int Cin, int Cout, int mode, const void* bias, const void* residual, int ldr,
                       float alpha, unsigned flags, float* gn_partial, int groups, void* stream);
/* 1 when the _gn launch of this shape gets its partial sums from the kernel's own epilogue (mode as in
 * skg_gemm_variant: 0 = GEMM, 1 + SKG_CONV_*), 0 when the stand-alone statistics pass follows. */
int skg_gemm_gn_fused(int M, int N, int K, int Cin, int mode, int HW, int groups);
int skg_groupnorm_from_partial(const void* X, int ldx, void* Y, int ldy, int rows, int HW, int C, int groups,
                               float eps, const void* gamma, const void* beta, int silu, float* stats,
                               const float* partial, int nch, void* stream);
/* The same for a CONCATENATION X = [A (CA channels) | B (C - CA channels)] whose halves were written by two producers
 * (the skip connection of the UNet's up path and the layer in front of it): partialA / partialB hold groupsA / groupsB
 * groups per chunk over each half's own channels; the concatenation's group width must be a multiple of both source
 * widths and CA a multiple of it (320 + 320, 640 + 640 channels: two source groups per output group), otherwise
 * SKG_E_UNSUPPORTED (nothing launched: run skg_groupnorm_fwd).  Replaces: the GroupNorm on torch.cat([h, skip]) in
 * diffusers' CrossAttnUpBlock2D / UpBlock2D resnets (modules/pipeline.py:96). */
int skg_groupnorm_from_partial2(const void* X, int ldx, void* Y, int ldy, int rows, int HW, int C, int CA,
                                int groups, float eps, const void* gamma, const void* beta, int silu,
                                float* stats, const float* partialA, int groupsA, const float* partialB,
                                int groupsB, int nch, void* stream);
/* C[(m / seg_rows) * seg_stride + m % seg_rows][n] = sum_k A[m][k] * B[n][k] + bias[n]: a GEMM over all batch rows whose output rows
 * land in per-batch-row slots of a longer buffer - clip_guided_attn concatenates [N image tokens ; 257 sketch tokens] per batch
 * row (modules/clip_guided_attn.py:111-119), so the image tokens' K / V of row b belong at rows [b L, b L + N) of the K / V
 * buffer; one launch instead of one per batch row.  M % seg_rows == 0, K % 64 == 0, C 16-byte aligned, ldc % 8 == 0. */
int skg_gemm_f16_rows(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                      const void* bias, int seg_rows, int seg_stride, void* stream);
/* Nearest-neighbour 2x upsample followed by a 3x3 convolution (diffusers Upsample2D: the up path of the UNet at
 * modules/pipeline.py:96, the VAE decoder at :118), POLYPHASE: out (2i+a, 2j+b) = sum over the 2 x 2 low-resolution pixels the
 * nine taps land on, with the taps that land on one pixel pre-summed - four 4-tap stride-1 convolutions over the low-res
 * input, 16 tap-products per input pixel instead of 36 (skg_conv3x3_f16 with SKG_CONV_UP2 is the 9-tap form, same result up
 * to the fp16 rounding of the summed weights).  X [rows*IH*IW, Cin] (ldx), Y [rows*2IH*2IW, Cout] (ldy), Wpp: packed
 * [4 phases 2a+b][Cout][4 taps][Cin] (sketch2img_amd.unet.pack_conv_up2).  Cin % 64 == 0. */
int skg_conv3x3_up2_f16(const void* X, int ldx, const void* Wpp, void* Y, int ldy, int rows, int IH, int IW, int Cin,
                        int Cout, const void* bias, void* stream);
/* Accuracy mode: the same on a pair input - X2 = [x_hi | x_lo], 2 C channels per pixel - with (hi, lo) pre-summed weights and a
 * pair output: per tap [x_hi | x_lo | x_hi] . [W_hi | W_hi | W_lo], Wpp3 [4 phases][Cout][4 taps][3 C]
 * (sketch2img_amd.unet.pack_conv_up2_hilo).  C % 64 == 0. */
int skg_conv3x3_up2_f16_hilo(const void* X2, int ldx, const void* Wpp3, void* Y, void* Y_lo, int ldy, int rows, int IH, int IW,
                             int C, int Cout, const void* bias, void* stream);
/* Accuracy mode, cheaper form (round 5): skg_conv3x3_up2_f16 on the hi part of the stream (X [rows*IH*IW][ldx >= Cin], Wpp the
 * default polyphase pack [4][Cout][4*Cin]) with the output written as the pair Y + Y_lo.  Drops the x_lo and W_lo correction
 * thirds of skg_conv3x3_up2_f16_hilo (each ~1 % of the mode's eps distance, together two thirds of its time). */
int skg_conv3x3_up2_f16_pairout(const void* X, int ldx, const void* Wpp, void* Y, void* Y_lo, int ldy, int rows, int IH, int IW,
                                int Cin, int Cout, const void* bias, void* stream);

/* conv2 + conv_shortcut of a ResnetBlock whose channel count changes, as ONE implicit GEMM (round 5; diffusers ResnetBlock2D:
 * output = conv2(.) + conv_shortcut(x), under modules/pipeline.py:96): the 1x1 shortcut is a tenth "tap" without halo -
 *   Y[m][n] = sum_{tap, c} X[pixel(m) + tap][c] * Wcat[n][tap * Cin + c] + sum_{c2 < K2} X2[m][c2] * Wcat[n][9 * Cin + c2] + bias[n]
 * X [rows*IH*IW][ldx >= Cin] the 3x3 input (stride 1), X2 [rows*IH*IW][ldx2 >= K2] the block's input x, Wcat [Cout][9*Cin + K2]
 * (conv2's tap-major pack followed by the shortcut weight; bias = conv2.bias + conv_shortcut.bias).  Y_lo != NULL (accuracy mode):
 * pair output, and X2 / K2 are the pair buffer [x_hi | x_lo] / 2 * Cin_x against [W_sc | W_sc].  gn_partial != NULL: the GroupNorm
 * partial sums of the output as skg_conv3x3_f16_gn writes them.  Removes one GEMM launch, its M x Cout output write and the
 * residual read of conv2's epilogue per such block (14 per SD1.5 evaluation).  Cin % 64 == 0, K2 % 64 == 0.  Returns
 * SKG_E_UNSUPPORTED (nothing launched: run conv2 and the shortcut GEMM as two launches) when X2 spans 2 GiB or more. */
int skg_conv3x3_sc_f16(const void* X, int ldx, const void* X2, int ldx2, int K2, const void* Wcat, void* Y, void* Y_lo, int ldy,
                       int rows, int IH, int IW, int Cin, int Cout, const void* bias, unsigned flags, float* gn_partial, int groups,
                       void* stream);
/* 3x3 stride-1 convolution (padding 1) by Winograd F(2x2, 3x3) - the 16 x 16-level ResnetBlock convolutions of the UNet evaluated at
 * modules/pipeline.py:96 and their data gradients inside torch.autograd.grad at modules/pipeline.py:159 (diffusers ResnetBlock2D conv1 /
 * conv2 -> cuDNN).  2.25 x fewer MFMA flops and 4 x more workgroups than the implicit GEMM on maps that under-fill the chip:
 *   V = B^T d B per 4 x 4 input tile (stride 2)  ->  16 GEMMs V_c [Mt][Cin] . U_c^T, Mt = rows * IH/2 * IW/2 (ONE split launch of the
 *   GEMM kernel, fp32 slabs in the stream's workspace)  ->  Y = A^T M A + bias (+ residual) per 2 x 2 output tile.
 * X [rows*IH*IW][ldx >= Cin] fp16; U [Cout][16 * Cin] fp16 = G g G^T per (cout, cin), component c = 4 i + j at columns [c Cin, (c + 1) Cin)
 * (unet.pack_conv_wino); V: caller-owned scratch of skg_conv3x3_wino_v_bytes() bytes; Y [rows*IH*IW][ldy] (+ Y_lo: pair output);
 * flags: SKG_EPI_RELU only.  Needs IH, IW even, Cin % 64 == 0, Cout % 8 == 0 and a registered workspace of >= 64 * Mt * Cout bytes
 * (skg_set_workspace); otherwise SKG_E_UNSUPPORTED (nothing launched: run skg_conv3x3_f16).  U and V carry one more fp16 rounding than
 * the implicit GEMM's operands (tools/eps_winograd.py prices it: default-mode eps rel 1.07e-3 -> 1.09e-3).
 * X == NULL: V already holds the input transform (skg_groupnorm_wino_fwd wrote it). */
size_t skg_conv3x3_wino_v_bytes(int rows, int IH, int IW, int Cin);
int skg_conv3x3_wino_f16(const void* X, int ldx, const void* U, void* V, void* Y, void* Y_lo, int ldy, int rows, int IH, int IW,
                         int Cin, int Cout, const void* bias, const void* residual, const void* residual_lo, int ldr, unsigned flags,
                         void* stream);
/* GroupNorm(+SiLU) of a small map whose consumer is skg_conv3x3_wino_f16: statistics published to `stats` as skg_groupnorm_fwd does,
 * and instead of the normalised tensor its Winograd input transform V [rows*IH/2*IW/2][16*C] (the fp16-rounded activations transformed:
 * bit-identical to skg_groupnorm_fwd + the convolution's own input transform; one launch and one [M, C] round trip less).  One workgroup
 * per (row, group) holds the slice in registers / LDS: C / groups a multiple of 8, IH * IW * C / groups / 8 <= 2560, even maps - else
 * SKG_E_UNSUPPORTED (nothing launched).  Same replaced reference calls as skg_groupnorm_fwd (ResnetBlock2D.norm1 / norm2). */
int skg_groupnorm_wino_fwd(const void* X, int ldx, void* V, int rows, int IH, int IW, int C, int groups, float eps,
                           const void* gamma, const void* beta, int silu, float* stats, void* stream);
/* Data gradient of the polyphase upsample + convolution above (the autograd backward of diffusers Upsample2D inside
 * torch.autograd.grad at modules/pipeline.py:159): ONE 4 x 4 stride-2 convolution, padding 1, over the gradient at the upsampled
 * size.  X [rows*IH*IW, Cin] (ldx; IH, IW even), Y [rows*(IH/2)*(IW/2), Cout] (ldy), W16 [Cout][16 taps ky*4+kx][Cin]
 * (sketch2img_amd.unet.pack_conv_up2_dgrad: the transposed pre-summed polyphase weights).  16 tap-products per output pixel
 * where the 9-tap dgrad at the upsampled size + 2 x 2 sum-pool spends 36.  Cin % 64 == 0. */
int skg_conv4x4s2_f16(const void* X, int ldx, const void* W16, void* Y, int ldy, int rows, int IH, int IW, int Cin,
                      int Cout, const void* bias, void* stream);

/* ---- accuracy mode ("residual_fp32"): tensors as (hi, lo) PAIRS of fp16, value = hi + lo (~22 mantissa bits) -------
 * north_star asks for <= 1e-3 max latent-eps deviation from the fp32 reference (modules/pipeline.py:96 in fp32 on CPU);
 * with every stored tensor in fp16 - what the reference's own GPU path does (app.py:34) - an evaluation is 1.5e-3 away,
 * 1.0e-3 of it from the fp16 residual stream.  HipUNet(residual_fp32=True) keeps the residual stream and the conv outputs
 * that feed a norm or the residual sum as pairs: the producer's epilogue writes hi = fp16(v) and lo = fp16(v - hi), adds a
 * residual pair in fp32; norms read hi + lo; where the stream itself is a matmul operand the pair is the operand
 * ([hi | lo] along K against [W | W]).  skg_gemm_f16 / skg_conv3x3_f16 with a second output / residual pointer (same leading
 * dimensions); C_lo / residual_lo may be NULL individually, not both.  fp16 output only, K % 64 == 0, 16-byte aligned. */
int skg_gemm_f16_hilo(const void* A, int lda, const void* B, int ldb, void* C, void* C_lo, int ldc, int M, int N, int K,
                      const void* bias, const void* residual, const void* residual_lo, int ldr, float alpha,
                      unsigned flags, void* stream);
int skg_conv3x3_f16_hilo(const void* X, int ldx, const void* Wp, void* Y, void* Y_lo, int ldy, int rows, int IH, int IW,
                         int Cin, int Cout, int mode, const void* bias, const void* residual, const void* residual_lo,
                         int ldr, float alpha, unsigned flags, void* stream);
/* The same launches leaving the GroupNorm partial sums of the output's hi part behind (format of skg_gemm_f16_gn /
 * skg_conv3x3_f16_gn): from the producer's own epilogue where its tile can (256 x 320, 128 x 160), else the stand-alone pass. */
int skg_gemm_f16_hilo_gn(const void* A, int lda, const void* B, int ldb, void* C, void* C_lo, int ldc, int M, int N, int K,
                         const void* bias, const void* residual, const void* residual_lo, int ldr, float alpha,
                         unsigned flags, float* gn_partial, int HW, int groups, void* stream);
int skg_conv3x3_f16_hilo_gn(const void* X, int ldx, const void* Wp, void* Y, void* Y_lo, int ldy, int rows, int IH, int IW,
                            int Cin, int Cout, int mode, const void* bias, const void* residual, const void* residual_lo,
                            int ldr, float alpha, unsigned flags, float* gn_partial, int groups, void* stream);
/* GroupNorm(+SiLU) of a pair X + X_lo (one pitch), mirrors of skg_groupnorm_fwd (own statistics pass over the hi part; small
 * maps: one launch on the pair's sum) and of skg_groupnorm_from_partial / _from_partial2 (partialB == NULL: one producer and
 * groupsA == groups; else the concatenation [A (CA channels) | B]).  The reference's GroupNorm layers (diffusers ResnetBlock2D /
 * Transformer2DModel under modules/pipeline.py:96) evaluated on the fp32-accurate stream.  Y_lo != NULL: the output is a pair too
 * (pitch ldy; conv_norm_out in front of conv_out, whose operand rounding would reach eps one to one). */
int skg_groupnorm_fwd_hilo(const void* X, const void* X_lo, int ldx, void* Y, void* Y_lo /* or NULL */, int ldy, int rows, int HW,
                           int C, int groups, float eps, const void* gamma, const void* beta, int silu, float* stats,
                           float* partial, void* stream);
int skg_groupnorm_from_partial_hilo(const void* X, const void* X_lo, int ldx, void* Y, void* Y_lo /* or NULL */, int ldy, int rows,
                                    int HW, int C, int CA,
                                    int groups, float eps, const void* gamma, const void* beta, int silu, float* stats,
                                    const float* partialA, int groupsA, const float* partialB, int groupsB, int nch,
                                    void* stream);
/* GroupNorm(+SiLU) apply and LayerNorm of a pair (statistics [rows][groups][2] from skg_groupnorm_stats on the hi part). */
int skg_groupnorm_apply_hilo(const void* X, const void* X_lo, int ldx, void* Y, int ldy, int rows, int HW, int C,
                             int groups, const float* stats, const void* gamma, const void* beta, int silu,
                             void* stream);
int skg_layernorm_fwd_hilo(const void* X, const void* X_lo, int ldx, void* Y, int ldy, int M, int C,
                           const void* gamma, const void* beta, float eps, float* stats /* [M][2] or NULL */, void* stream);

int skg_groupnorm_apply(const void* X, int ldx, void* Y, int ldy, int rows, int HW, int C,
                        int groups, const float* stats, const void* gamma, const void* beta,
                        int silu, void* stream);
int skg_groupnorm_bwd(const void* X, int ldx, const void* dY, int lddy, void* dX, int lddx,
                      const void* residual, int ldr, int rows, int HW, int C, int groups,
                      const float* stats, const void* gamma, const void* beta, int silu,
                      float* partial, void* stream);

/* ---- LayerNorm over the last dim, fp16 [M][C] ------------------------------------------------
 * fwd also writes float2 (mean, rstd) per row to `stats` if non-NULL.
 * bwd: dX = LN'(dY) + residual.  Requires C % 8 == 0, C <= 2048.
 * Replaces: BasicTransformerBlock.norm1/2/3 and sketch_norm (modules/clip_guided_attn.py:113). */
int skg_layernorm_fwd(const void* X, int ldx, void* Y, int ldy, int M, int C, const void* gamma,
                      const void* beta, float eps, float* stats, void* stream);
int skg_layernorm_bwd(const void* X, int ldx, const void* dY, int lddy, void* dX, int lddx,
                      const void* residual, int ldr, int M, int C, const void* gamma,
                      const float* stats, void* stream);

/* skg_gemm_f16 with SKG_EPI_GEGLU that ALSO stores the pre-activation: Y [M][N/2] = a * gelu(g) and H [M][N] = A B^T + bias
 * in the interleaved pack order (what skg_geglu_bwd takes as its saved H).  One launch instead of GEMM + skg_geglu_fwd
 * for the rows whose gate will be differentiated (the cond rows of a guided step).  K % 64 == 0, N % 16 == 0. */
int skg_gemm_f16_geglu_keep(const void* A, int lda, const void* B, int ldb, void* Y, int ldy, void* H, int ldh,
                            int M, int N, int K, const void* bias, void* stream);

/* ---- Row-local fused feed-forward sub-block (round 3) -----------------------------------
 * Y [M][C] = X + b2 + W2 . geglu(W1 . LayerNorm(X; gamma, beta, eps) + b1)   - ONE launch instead of skg_layernorm_fwd,
 * skg_gemm_f16 with SKG_EPI_GEGLU and skg_gemm_f16 with a residual; the [M][F] gated tensor never exists in memory.
 * C == 320 (the 64 x 64 level of SD1.5, the 96 x 96 level of SD2.1), F = hidden width (1280), F % 32 == 0, F <= 1280.
 * Wpack: fp16 [F/32][60][512] "fragment-major" pack of W1 [2F][C] (diffusers order: value rows, then gate rows) and
 * W2 [C][F]: chunk c holds the 32 hidden units 32c..32c+31 as 40 W1 pieces (tile t = val 0-15, val 16-31, gate 0-15,
 * gate 16-31; k-step ks; piece [lane = 16 g + l][8] = W1[row(t, l)][32 ks + 8 g ..]) then 20 W2 pieces (output tile u;
 * piece [lane][i] = W2[16 u + l][32 c + 16 (i >> 2) + 4 g + (i & 3)]); bias1_pack: fp32 [F/32][4][16] in W1-tile order
 * (sketch2img_amd.unet.pack_ff_block builds both).  stats (optional): float2 (mean, rstd) per row, as skg_layernorm_fwd.
 * Y may alias X.  Same roundings as the three-launch path (fp16 LayerNorm output, fp16 FF1 output before the gate,
 * fp16 gated value, one fp16 rounding of the residual sum).
 * Replaces: BasicTransformerBlock.norm3 / ff (GEGLU) / residual add ([diffusers] attention.py), reached from
 * modules/pipeline.py:96. */
int skg_ff_block_f16(const void* X, int ldx, void* Y, int ldy, int M, int C, int F, const void* gamma,
                     const void* beta, float eps, const void* Wpack, const float* bias1_pack,
                     const void* bias2, float* stats, void* stream);
/* The same launch, ALSO storing the FF1 output H = fp16(W1 . LayerNorm(X) + b1) of rows >= keep_from to H [M - keep_from][2F]
 * (row m -> H row m - keep_from) in the interleaved pack order skg_geglu_bwd takes as its saved H (what
 * skg_gemm_f16_geglu_keep writes): the cond rows of a guided step, whose gate is differentiated.  keep_from % 16 == 0,
 * ldh % 8 == 0.  H == NULL: plain skg_ff_block_f16. */
int skg_ff_block_f16_keep(const void* X, int ldx, void* Y, int ldy, int M, int C, int F, const void* gamma,
                          const void* beta, float eps, const void* Wpack, const float* bias1_pack,
                          const void* bias2, float* stats, void* H, int ldh, int keep_from, void* stream);
/* ... followed by Transformer2DModel.proj_out + the transformer's outer residual in the same launch ([diffusers] attention.py
 * Transformer2DModel.forward: `hidden_states = self.proj_out(hidden_states); output = hidden_states + residual`):
 *   Y = R + bias_proj + W_proj . fp16(X + FF(LayerNorm(X)))
 * Wpack holds five more chunks behind the F / 32 feed-forward ones (sketch2img_amd.unet.pack_ff_block(..., w_proj)); the block
 * output never reaches memory (its backward does not read it).  stats / H / ldh / keep_from as skg_ff_block_f16_keep (H == NULL:
 * no stash); gn_partial != NULL: also the GroupNorm partial sums of Y, fp32 [M / HW][HW / 128][groups][2], the layout of
 * skg_gemm_f16_gn (HW % 128 == 0, groups <= 32).  Y must not alias X; it may alias R. */
int skg_ff_block_proj_f16(const void* X, int ldx, void* Y, int ldy, int M, int C, int F, const void* gamma, const void* beta,
                          float eps, const void* Wpack, const float* bias1_pack, const void* bias2, const void* bias_proj,
                          const void* R, int ldr, float* stats, void* H, int ldh, int keep_from, float* gn_partial, int HW,
                          int groups, void* stream);

/* Accuracy mode (round 5): skg_ff_block_proj_f16 on pairs - X + X_lo in (pitch ldx), Y + Y_lo out (pitch ldy), outer residual
 * R + R_lo (pitch ldr).  proj_out takes the block output as the pair it is, W_proj . hi + W_proj . lo on the same weight
 * fragments (the K-doubled [p3_hi | p3_lo] . [W | W] GEMM of the unfused accuracy-mode path, without its second weight read and
 * without the [M, 2C] round trip of p3); gn_partial: the GroupNorm partial sums of Y's hi part.  Y must not alias X. */
int skg_ff_block_proj_f16_hilo(const void* X, const void* X_lo, int ldx, void* Y, void* Y_lo, int ldy, int M, int C, int F,
                               const void* gamma, const void* beta, float eps, const void* Wpack, const float* bias1_pack,
                               const void* bias2, const void* bias_proj, const void* R, const void* R_lo, int ldr, float* stats,
                               void* H, int ldh, int keep_from, float* gn_partial, int HW, int groups, void* stream);
/* Accuracy mode: the same launch on a PAIR input X + X_lo (pitch ldx) with a PAIR output Y + Y_lo (pitch ldy): LayerNorm reads
 * the sum, the residual sum is formed in fp32 and stored as hi = fp16(v), lo = fp16(v - hi).  H / keep_from as _keep (or NULL). */
int skg_ff_block_f16_hilo(const void* X, const void* X_lo, int ldx, void* Y, void* Y_lo, int ldy, int M, int C, int F,
                          const void* gamma, const void* beta, float eps, const void* Wpack, const float* bias1_pack,
                          const void* bias2, float* stats, void* H, int ldh, int keep_from, void* stream);

/* ---- Row-local fused cross-attention sub-block (round 3; VERDICT r2 next #2, first half) ---------------------------
 * Y [M][C] = X + bo + Wo . Attention(Q = Wq . LayerNorm(X; gamma, beta, eps), K, V) over the Nkv <= 80 text keys of the row's
 * image - ONE launch instead of skg_layernorm_fwd, skg_gemm_f16 (attn2.to_q), skg_attn_fwd_rowv (77 keys) and skg_gemm_f16 with a
 * residual (attn2.to_out); q, the attention output and the normalised rows never exist in memory.
 * C == 320, heads == 8 (head width 40: the 64 x 64 level of SD1.5); rows [b * HW, (b + 1) * HW) belong to image b, HW % 128 == 0,
 * M % HW == 0; scale = head_width^-0.5.
 * Wpack: fp16 [heads][60][512], KVpack: fp16 [M / HW][heads][16][512] - fragment-major images of Wq / Wo and of the per-image
 * text keys / values (layouts: sketch2img_amd.unet.pack_xattn_weights / pack_xattn_kv, which build them; KVpack once per prompt).
 * Same rounding points as the four-launch path (fp16 LayerNorm output, fp16 q, fp16 scaled q, fp16 probabilities, fp16
 * attention output, one fp16 rounding of the residual sum).  Y may alias X.  No LSE / q / o outputs: for rows nobody
 * differentiates (unguided evaluations, the uncond half of guided ones).
 * Replaces: BasicTransformerBlock.norm2 / attn2 / residual add ([diffusers] attention.py; the op order of
 * modules/clip_guided_attn.py:127-152), reached from modules/pipeline.py:96. */
int skg_xattn_block_f16(const void* X, int ldx, void* Y, int ldy, int M, int HW, int C, int heads, int Nkv,
                        const void* gamma, const void* beta, float eps, const void* Wpack, const void* KVpack,
                        const void* bias_out, float scale, void* stream);
/* The stashing launch of a guided evaluation (as skg_ff_block_f16_keep): rows >= keep_from - a multiple of HW, i.e. whole images:
 * the cond half - ALSO store what the backward of the four replaced launches reads: stats fp32 [M - keep_from][2] = norm2's
 * (mean, rstd) per row, Q / O fp16 [M - keep_from][ldk] = the attn2.to_q output and the attention output (head h in columns
 * 40 h .. 40 h + 39), lse fp32 [(M - keep_from) / HW][heads][HW] in skg_attn_fwd's convention.  Feeds skg_attn_bwd_delta /
 * skg_attn_bwd_dq / skg_layernorm_bwd unchanged. */
int skg_xattn_block_f16_keep(const void* X, int ldx, void* Y, int ldy, int M, int HW, int C, int heads, int Nkv,
                             const void* gamma, const void* beta, float eps, const void* Wpack, const void* KVpack,
                             const void* bias_out, float scale, float* stats, void* Q, void* O, int ldk, float* lse,
                             int keep_from, void* stream);
/* ... on pairs (accuracy mode, round 5): skg_xattn_block_f16_hilo that also stores, for the rows >= keep_from, what the backward of
 * the replaced launches reads (stats, Q, O, lse as skg_xattn_block_f16_keep). */
int skg_xattn_block_f16_hilo_keep(const void* X, const void* X_lo, int ldx, void* Y, void* Y_lo, int ldy, int M, int HW, int C,
                                  int heads, int Nkv, const void* gamma, const void* beta, float eps, const void* Wpack,
                                  const void* KVpack, const void* bias_out, float scale, float* stats, void* Q, void* O, int ldk,
                                  float* lse, int keep_from, void* stream);
/* Accuracy mode: the same launch on a PAIR input X + X_lo (pitch ldx) with a PAIR output Y + Y_lo (pitch ldy). */
int skg_xattn_block_f16_hilo(const void* X, const void* X_lo, int ldx, void* Y, void* Y_lo, int ldy, int M, int HW, int C,
                             int heads, int Nkv, const void* gamma, const void* beta, float eps, const void* Wpack,
                             const void* KVpack, const void* bias_out, float scale, void* stream);

/* ---- GEGLU: Y[m][j] = a_j * gelu(g_j),  H fp16 [M][2F] -----------------------------------------------
 * interleaved == 0: H = [a (F columns) | g (F columns)] (diffusers' chunk(2));  interleaved == 1: groups of four
 * columns [a_2t a_2t+1 g_2t g_2t+1] (the pack SKG_EPI_GEGLU uses).  bwd writes dH [M][2F] in the same layout from
 * dY [M][F] and the saved H.  F % 8 == 0.
 * Replaces: diffusers GEGLU (ff.net.0) inside BasicTransformerBlock. */
int skg_geglu_fwd(const void* H, int ldh, void* Y, int ldy, int M, int F, int interleaved, void* stream);
int skg_geglu_bwd(const void* H, int ldh, const void* dY, int lddy, void* dH, int lddh, int M,
                  int F, int interleaved, void* stream);

/* ---- fused multi-head attention (flash style), fp16 --------------------------------------------
 * Q [batch*Nq][..] (ldq), K [batch*kv_stride][..] (ldk), head h occupies columns
 * [h*dh, (h+1)*dh).  Vt is V transposed: Vt[h*dh + d][b*kv_stride + j] (ldvt).  Only the first
 * Nkv of each batch row's kv_stride key slots are valid.  O [batch*Nq][..] (ldo).
 * lse (float [batch][heads][Nq], natural-log-sum-exp of the scaled scores) may be NULL.
 * dh in {16, 32, 40, 64, 80, 160}; kv_stride % 8 == 0.
 * Replaces: xformers memory_efficient_attention (enabled at app.py:43) / diffusers
 * CrossAttention baddbmm+softmax+bmm for attn1, attn2 and the injected sketch_attn
 * (modules/clip_guided_attn.py:114, modules/sketch_guided_attn.py:127). */
int skg_attn_fwd(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt,
                 void* O, int ldo, float* lse, int batch, int heads, int Nq, int Nkv,
                 int kv_stride, int dh, float scale, void* stream);
/* Same with V handed over ROW-MAJOR: V [batch*kv_stride][ldv], head h at columns h*dh.. (e.g. the third column block
 * of a fused QKV projection, ldv = 3C).  The kernel reads its V fragments through the gfx950 LDS transpose read
 * (ds_read_b64_tr_b16), so no transposed copy of V is written.  Same replaced reference calls as skg_attn_fwd. */
int skg_attn_fwd_rowv(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                      void* O, int ldo, float* lse, int batch, int heads, int Nq, int Nkv,
                      int kv_stride, int dh, float scale, void* stream);
/* Same with a causal mask: key j of a batch row is visible to query i only when j <= i (self-attention, Nq rows
 * and kv_stride key slots per batch row).  dh in {16, 32, 64}.
 * Replaces: the masked self-attention of transformers' CLIPTextModel, which the reference calls through
 * StableDiffusionPipeline._encode_prompt (modules/pipeline.py:55-57). */
int skg_attn_fwd_causal(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt,
                        void* O, int ldo, float* lse, int batch, int heads, int Nq, int Nkv,
                        int kv_stride, int dh, float scale, void* stream);
/* backward.  delta[b][h][q] = sum_d dO*O (skg_attn_bwd_delta).  All operands row-major like their primals (Q, dO
 * [batch*Nq][..], K, V [batch*kv_stride][..]); the transposed fragments the dQ / dK / dV products need (K^T, Q^T, dO^T)
 * are read from the row tiles in LDS through ds_read_b64_tr_b16 - no transposed copies in HBM.  Outputs row-major.
 * Replaces: torch autograd through xformers memory_efficient_attention / CrossAttention, triggered by
 * torch.autograd.grad at modules/pipeline.py:159. */
int skg_attn_bwd_delta(const void* O, int ldo, const void* dO, int lddo, float* delta, int batch,
                       int heads, int Nq, int dh, void* stream);
int skg_attn_bwd_dq(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                    const void* dO, int lddo, const float* lse,
                    const float* delta, void* dQ, int lddq, int batch, int heads, int Nq, int Nkv,
                    int kv_stride, int dh, float scale, void* stream);
/* skg_attn_bwd_dq with skg_attn_bwd_delta in its prologue (round 5): O [batch*Nq][ldo] is the forward's output; delta is formed from
 * the dO fragments the launch loads anyway, used, and stored to delta_out [batch][heads][Nq] for the skg_attn_bwd_dkv launch
 * that follows (summation order differs from skg_attn_bwd_delta's: equal to fp32 rounding, not bit for bit). */
int skg_attn_bwd_dq_delta(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* dO, int lddo,
                          const void* O, int ldo, const float* lse, float* delta_out, void* dQ, int lddq, int batch, int heads,
                          int Nq, int Nkv, int kv_stride, int dh, float scale, void* stream);
int skg_attn_bwd_dkv(const void* Q, int ldq, const void* K, int ldk,
                     const void* V, int ldv, const void* dO, int lddo,
                     const float* lse, const float* delta, void* dK, int lddk, void* dV, int lddv,
                     int batch, int heads, int Nq, int Nkv, int dh, float scale, void* stream);

/* ---- data movement -----------------------------------------------------------------------------*/
/* Out[c][m] = In[m][c], fp16, In [M][C] (ldi), Out [C][M] (ldo).  C % 8 == 0, M % 8 == 0. */
int skg_transpose_f16(const void* In, int ldi, void* Out, int ldo, int M, int C, void* stream);
/* Y[m][c] = alpha*A[m][c] + beta*B[m][c] (B may be NULL); fp16, C % 8 == 0.  Used for channel
 * concat (copy into a slice), skip-gradient accumulation and slicing. */
int skg_axpby_f16(const void* A, int lda, const void* B, int ldb, void* Y, int ldy, int M, int C,
                  float alpha, float beta, void* stream);
/* Out[b*out_batch_rows + r][:] = In[b*in_batch_rows + r][:] for b < batches, r < rows_per_batch (fp16, C % 8 == 0).
 * Moves the N image tokens of every batch row into / out of the [N + 257 (+pad)]-token buffer of the
 * injected self-attention, i.e. the torch.cat(..., dim=1) and [:, :N] slice of
 * modules/clip_guided_attn.py:113,119. */
int skg_batch_copy_f16(const void* In, int ldi, int in_batch_rows, void* Out, int ldo, int out_batch_rows,
                       int batches, int rows_per_batch, int C, void* stream);
/* Y = silu(X), fp16 [M][C], C % 8 == 0.  Used once per timestep for the time-embedding MLP
 * (diffusers TimestepEmbedding / ResnetBlock2D.time_emb_proj input), off the per-step path. */
int skg_silu_f16(const void* X, int ldx, void* Y, int ldy, int M, int C, void* stream);
/* Y = X * sigmoid(1.702 X): transformers' "quick_gelu", the MLP activation of the CLIP vision tower that
 * modules/clip_guided_inf.py:49-54,103 runs to produce the sketch tokens. */
int skg_quick_gelu_f16(const void* X, int ldx, void* Y, int ldy, int M, int C, void* stream);
/* Y = X/2 (1 + erf(X / sqrt 2)): exact "gelu", the MLP activation of the OpenCLIP text encoder that SD 2.x
 * checkpoints carry (transformers CLIPTextModel with hidden_act = "gelu"; modules/pipeline.py:55-57). */
int skg_gelu_f16(const void* X, int ldx, void* Y, int ldy, int M, int C, void* stream);
/* adjoint of nearest 2x upsample: Y[b][y][x][c] = sum of the 2x2 block of X.  X [rows*2H*2W][C]. */
int skg_sumpool2x2_f16(const void* X, int ldx, void* Y, int ldy, int rows, int H, int W, int C,
                       void* stream);
/* NCHW float [rows][C][HW]  ->  NHWC fp16 [rows*HW][Cpad] (channels >= C zero-filled) and back
 * (the inverse reads the first C channels of fp16 [rows*HW][ldx] into float NCHW). */
int skg_nchw_f32_to_nhwc_f16(const float* X, void* Y, int rows, int C, int HW, int Cpad, void* stream);
int skg_nhwc_f16_to_nchw_f32(const void* X, int ldx, float* Y, int rows, int C, int HW, void* stream);

/* ---- LGP (Latent Gradient/Edge Predictor) pieces ------------------------------------------------
 * Layer 0 is re-associated with the bilinear resize: per tap i the caller runs
 * skg_gemm_f16(F_i, W0[:, off_i:off_i+C_i]) -> P_i fp32 [rows*s_i*s_i][H0] at the tap's NATIVE
 * size; skg_lgp_layer0_gather then forms, for every output pixel of the h x h grid,
 *   z = relu( fp16( sum_i bilinear_i(P_i) + W0[:, E:E+40] . fp16(e) + b0 ) )
 * with e = [nl(4), sin(2 pi nl 2^-l) l=0..8 (36)], nl = sigma * noise  (fp32, then fp16 cast, as
 * modules/latent_predictor.py:39-43).  Output Z fp16 [rows*h*h][H0], pixel order (y, x).
 * `taps` is a HOST array of ntaps (<= 12) SkgLgpTap; rows = 2*samples in the row layout
 * [uncond rows of all samples; cond rows of all samples]; noise float NCHW [samples][4][h][h].  Replaces F.interpolate x9 + cat + cast + layer 0 of
 * modules/pipeline.py:146-153 / modules/latent_predictor.py:42-45. */
typedef struct {
  const float* P;   /* fp32 [rows*s*s][H0] */
  int s;            /* native side */
  int pad_;
} SkgLgpTap;
/* Wextra may be NULL (h % 8 == 0 and H0 % 128 == 0 required): the caller then supplies the product of the 40 extra
 * channels (skg_lgp_extra_features x W0[:, E:], an fp32 GEMM) as one more tap with s == h - the matrix pipe does the
 * 40-wide dot products instead of 160 cached vector loads per thread. */
int skg_lgp_layer0_gather(const SkgLgpTap* taps, int ntaps, const void* Wextra, int ldw,
                          const void* bias0, const float* noise, float sigma, int samples,
                          void* Z, int rows, int h, int H0, void* stream);
/* adjoint of the bilinear part for ONE tap: dP [rows*s*s][H0] fp16 from dZ [rows*h*h][H0] fp16. */
int skg_lgp_layer0_scatter(const void* dZ, int lddz, void* dP, int rows, int h, int s, int H0,
                           void* stream);
/* train-mode BatchNorm1d whose "batch" is ONE sample's LGP rows (the reference only runs B = 1:
 * SURVEY Q1/Q3).  Row layout of X: row = (j*samples + s)*seg_rows + i, j = 0..segs-1 (segs = 2 CFG
 * halves, seg_rows = h*h).  X is the post-ReLU activation fp16 [samples*segs*seg_rows][C].
 * stats float [samples][C][2] = (mean, rstd), biased variance.  If running_mean/var != NULL
 * (float [C]) they receive the momentum-0.1 / unbiased-variance update once per sample, in sample
 * order (= the side effect of `samples` consecutive reference calls).
 * apply:    Y = fp16((x-mean)*rstd*gamma+beta).
 * relu_bwd: dX = [x>0] * gamma*rstd*(dY - mean_s(dY) - xhat*mean_s(dY*xhat))  - BatchNorm backward
 *           including the statistic terms, then the ReLU that precedes the BN in
 *           modules/latent_predictor.py:15-27 (x is post-ReLU so relu'(.) = [x>0]).  With
 *           train_mode == 0 the statistic terms are dropped (eval-mode BN: stats from running).
 * scratch: skg_bn_scratch_floats(samples, C) floats. */
size_t skg_bn_scratch_floats(int samples, int C);
int skg_bn_stats(const void* X, int ldx, int samples, int segs, int seg_rows, int C, float eps,
                 float* stats, float* scratch, float* running_mean, float* running_var,
                 void* stream);
int skg_bn_stats_from_running(const float* running_mean, const float* running_var, int samples,
                              int C, float eps, float* stats, void* stream);
int skg_bn_apply(const void* X, int ldx, void* Y, int ldy, int samples, int segs, int seg_rows,
                 int C, const float* stats, const void* gamma, const void* beta, void* stream);
int skg_bn_relu_bwd(const void* X, int ldx, const void* dY, int lddy, void* dX, int lddx,
                    int samples, int segs, int seg_rows, int C, const float* stats,
                    const void* gamma, int train_mode, float* scratch, void* stream);
/* MSE seed: for the cond rows, dOut = loss_scale * 2*(out - target)/(n) with n = 4*h*h, zero for the
 * uncond rows; also writes loss (float[samples]).  out fp16 [2*samples*h*h][ldo] (uncond block then
 * cond block, (y,x) pixel order), first 4 columns valid; target float NCHW [samples][4][h][h].  dOut fp16 [..][ldd],
 * columns >= 4 zeroed up to ldd.  Mirrors modules/pipeline.py:155-157. */
int skg_lgp_mse_seed(const void* out, int ldo, const float* target, void* dOut, int ldd,
                     float* loss, int samples, int h, float loss_scale, void* stream);

/* ---- LGP training (trainer.py:208-252: forward taps -> LGP -> MSE -> backward with weight gradients -> optimizer) --
 * The reference trains with accelerate fp16 autocast + bitsandbytes AdamW8bit; here: fp16 compute with a static loss
 * scale, fp32 master weights, plain AdamW.  Weight gradients are GEMMs on transposed operands (skg_transpose_f16 +
 * skg_gemm_f16 with SKG_EPI_OUT_F32); these are the remaining pieces. */
/* out[c] = scale * sum_m X[m][c]  (bias gradients).  scratch: skg_colsum_scratch_floats(C) floats. */
size_t skg_colsum_scratch_floats(int C);
int skg_colsum_f16(const void* X, int ldx, int M, int C, float scale, float* out, float* scratch, void* stream);
/* BatchNorm1d parameter gradients over `rows` rows of ONE batch: dgamma[c] = scale * sum dY*xhat, dbeta[c] = scale *
 * sum dY, xhat from stats (mean, rstd)[C] of skg_bn_stats(samples = 1).  scratch: skg_bn_scratch_floats(1, C). */
int skg_bn_param_grads(const void* X, int ldx, const void* dY, int lddy, int rows, int C, const float* stats,
                       float scale, float* dgamma, float* dbeta, float* scratch, void* stream);
/* The 40 extra input channels of LGP layer 0 (noise level x4, sin(2 pi nl 2^-l) x36; modules/latent_predictor.py:39-41),
 * fp16, E [rows*h*h][ld] with columns >= 40 zeroed: the operand of the layer-0 weight gradient for those columns. */
int skg_lgp_extra_features(const float* noise, float sigma, int samples, int rows, int h, void* E, int ld,
                           void* stream);
/* Training loss (trainer.py:240): mean((out - target)^2) over samples*4*h*h; dOut = loss_scale * d loss / d out
 * (fp16 [samples*h*h][ldd], columns >= 4 zero); loss_parts[s] = this sample's share of the mean (sum them). */
int skg_lgp_mse_train(const void* out, int ldo, const float* target, void* dOut, int ldd, float* loss_parts,
                      int samples, int h, float loss_scale, void* stream);
/* One AdamW step on a flat fp32 parameter vector (decoupled weight decay, bias correction for `step` >= 1); grad is
 * multiplied by inv_grad_scale first; param_f16 (optional) receives the fp16 working copy. */
int skg_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_f16, size_t n,
                   float lr, float beta1, float beta2, float eps, float weight_decay, int step, float inv_grad_scale,
                   void* stream);

/* ---- sampler elementwise -------------------------------------------------------------------------
 * CFG combine + DDIM step (eta = 0) on float NCHW latents:
 *   eps = eps_u + g*(eps_c - eps_u);  x0 = (x - c1*eps)/c0;  x_prev = c2*x0 + c3*eps
 * eps_u / eps_c are fp16 NHWC [HW][ld] rows of the UNet output (first 4 channels); lo_off != 0 (accuracy mode): eps is a
 * (hi, lo) pair whose lo part sits lo_off columns to the right in the same rows (0 = plain fp16).
 * vpred = 1 (scheduler config prediction_type "v_prediction", the public SD2.1-768 checkpoint): the UNet output is v;
 *   v = v_u + g*(v_c - v_u);  eps = c0*v + c1*x;  x0 = c0*x - c1*v;  x_prev as above (eps_out receives the derived eps).
 * Replaces modules/pipeline.py:99-104 (CFG + scheduler.step). */
int skg_cfg_ddim_step(const void* eps_u, const void* eps_c, int ld, int lo_off, const float* x, float* x_prev,
                      float* eps_out, int samples, int HW, float g, float c0, float c1, float c2,
                      float c3, int vpred, void* stream);
/* ---- VAE decoder helpers (modules/pipeline.py:118 decode_latents; third-party AutoencoderKL.decode) -------------
 * Row softmax y[m][:] = softmax(x[m][:]) of fp16 scores, fp32 arithmetic: the decoder's single-head mid attention
 * (AttentionBlock: softmax(q k^T / sqrt(C)) v over HW tokens) is GEMM -> this -> GEMM.  N % 8 == 0. */
int skg_softmax_rows_f16(const void* x, int ldx, void* y, int ldy, int M, int N, void* stream);
/* decode_latents tail: out[px][c] = clamp(x[px][c]*scale + shift, 0, 1), fp16 NHWC (row pitch ld) -> float NHWC
 * (the (B, H, W, 3) array the pipeline converts to PIL; scale 0.5, shift 0.5). */
int skg_image_postprocess(const void* x, int ld, float* out, size_t pixels, int C, float scale, float shift,
                          void* stream);

/* The same tail fused with numpy_to_pil's quantisation (modules/pipeline.py:125 -> diffusers numpy_to_pil:
 * (images * 255).round().astype("uint8")): out[px][c] = rint(clamp(x*scale + shift, 0, 1) * 255), uint8 NHWC
 * [pixels][C] - the (S, 512, 512, 3) payload a rank contributes to the final gather of decoded images. */
int skg_image_to_u8(const void* x, int ld, void* out, size_t pixels, int C, float scale, float shift, void* stream);

/* VAE encoder tail (app.py:109: vae.encode(img).latent_dist.sample() * 0.18215): from the fp16 NHWC moments
 * [samples*HW][ld] = (mean[0..L), logvar[L..2L)) to float NCHW [samples][L][HW]:
 *   out = (mean + exp(0.5 * clamp(logvar, -30, 20)) * noise) * scale;   noise NULL -> the mode (mean * scale).
 * noise is a caller-drawn standard normal tensor in the output layout (the reference draws it from torch's RNG). */
int skg_gaussian_sample(const void* moments, int ld, const float* noise, float* out, int samples, int L, int HW,
                        float scale, void* stream);

/* CFG combine + one DPM-Solver++ (2M, midpoint) update - the scheduler app.py:13-25 configures - on float NCHW latents:
 *   eps = eps_u + g*(eps_c - eps_u);  x0 = (x - sigma_s*eps)/alpha_s;  x_prev = a*x + b*x0 + c*x0_before
 * x0_io [samples][4][HW] float: the previous step's x0 on entry (not read when c == 0: first-order step), this
 * step's x0 on exit.  The five scalars are host-side fp32 table arithmetic (sketch2img_amd/sampler.py DPMTables).
 * vpred = 1: the UNet output is v: x0 = alpha_s*x - sigma_s*v (eps_out: alpha_s*v + sigma_s*x).
 * Replaces modules/pipeline.py:99-104 when the pipeline was built with DPMSolverMultistepScheduler. */
int skg_cfg_dpmpp2m_step(const void* eps_u, const void* eps_c, int ld, int lo_off, const float* x, float* x0_io,
                         float* x_prev, float* eps_out, int samples, int HW, float g, float alpha_s,
                         float sigma_s, float a, float b, float c, int vpred, void* stream);
/* guidance update, modules/pipeline.py:159-161, per sample s:
 *   g = -grad[s] (fp16 NHWC [HW][ld], first 4 ch);  alpha = sqrt(2)*||x_in - x_prev|| / ||g|| * beta
 *   x_prev += alpha * g.   aux float [samples][4] receives (alpha, ||g||, ||x_in-x_prev||*sqrt2, 0). */
int skg_guidance_update(const void* grad, int ld, const float* x_in, float* x_prev, float* aux,
                        int samples, int HW, float beta, void* stream);

/* Box calibration (bench.py `config.box_mfma_tflops`; no reference call site - measurement infrastructure): one launch of
 * SKG_BOX_PROBE_WORKGROUPS workgroups x 8 waves, each wave `iters` rounds of 40 register-resident 16x16x32 fp16 MFMAs on
 * pseudo-random operands = SKG_BOX_PROBE_WORKGROUPS * 8 * iters * 40 * 16384 flop.  out: float [SKG_BOX_PROBE_WORKGROUPS * 512]
 * (the accumulator sums: only there so that the loop is not removed).  Time it with events on `stream`. */
#define SKG_BOX_PROBE_WORKGROUPS 512
int skg_box_probe_mfma(float* out, int iters, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SKG_H_ */

// Parameter block shared by the GEMM / implicit-GEMM conv kernels (gemm.hip: generic v1, gemm2.hip: v2).
#pragma once
#include "common.h"

enum { MODE_DIRECT = 0, MODE_S1 = 1, MODE_S2 = 2, MODE_UP2 = 3, MODE_S2T = 4, MODE_S2A = 5 };

struct GemmParams {
  const half_t* A; int lda;
  const half_t* B; int ldb;
  void* C; int ldc;
  const half_t* bias;
  const half_t* res; int ldr;
  int M, N, K;
  float alpha; unsigned flags;
  int IH, IW, OH, OW, Cin;   // conv only
  // GroupNorm statistics of the OUTPUT from the producer's epilogue (skg_*_gn entry points): per (sample, 128-row chunk,
  // group) sum(y) and sum(y^2) of the fp16-rounded outputs, gn_partial[((b * (gn_hw / 128) + chunk) * gn_groups + g) * 2]
  float* gn_partial; int gn_hw, gn_groups;
  // fused GEGLU that also keeps the pre-activation (skg_gemm_f16_geglu_keep): H [M][N] in the interleaved pack order
  half_t* aux; int ldaux;
  // accuracy mode (skg_*_hilo entry points): the residual and / or the output are PAIRS of fp16 tensors whose sum carries
  // ~22 mantissa bits - lo = fp16(v - fp16(v)); same leading dimension as the hi part (ldr / ldc)
  const half_t* res_lo; half_t* c_lo;
  // polyphase nearest-2x-upsample + 3x3 conv (skg_conv3x3_up2_f16): a stride-1 conv over the LOW-resolution input that walks
  // only `ntaps` of the nine taps (tap id of walk position i = (tapmap >> 4 i) & 15; weight pack [Cout][ntaps][Cin]) and stores
  // low-res pixel (i, j) to high-res pixel (2 i + a, 2 j + b) of the output, up2 = 1 + 2 a + b (0 = ordinary conv)
  int ntaps; unsigned tapmap; int up2;
  // accuracy-mode polyphase upsample (skg_conv3x3_up2_f16_hilo): the K axis of a tap is [x_hi | x_lo | x_hi] against
  // [W_hi | W_hi | W_lo] but the operand only holds [x_hi | x_lo]: channel offsets >= a_wrap wrap around to the start (0 = off)
  int a_wrap;
  // segmented output rows (skg_gemm_f16_rows): row m of the product is stored to row (m / seg_rows) * seg_stride + m % seg_rows
  // of C - one launch writes the image tokens of every batch row into its slot of a longer per-row buffer (0 = off)
  int seg_rows, seg_stride;
  // conv2 + conv_shortcut as ONE implicit GEMM (skg_conv3x3_sc_f16, MODE_S1 only, round 5): K tiles past the 9 * Cin of the 3x3
  // walk read a SECOND row-major operand A2 [M][lda2 >= K2] (the ResnetBlock's input x: the 1x1 shortcut as a tenth "tap"
  // without halo) against columns [9 * Cin, 9 * Cin + K2) of the weight rows; K = 9 * Cin + K2 (0 = off)
  const half_t* A2; int lda2, K2;
  // Winograd F(2x2, 3x3) GEMM step (wino.hip, round 6): a MODE_DIRECT launch on V [Mt][16 Cin] x U [N][16 Cin] whose K range is cut into
  // exactly `wino` (= 16) slices, slab c = transform component c; the launcher writes the raw fp32 slabs to p.ws and launches NO
  // reduce (the caller's output transform reads them) (0 = off)
  int wino;
  // split-K workspace of the launch stream (host side: filled by the entry points from the per-stream registry of
  // skg_set_workspace; the kernels get the slab pointer as an argument)
  float* ws; size_t ws_bytes;
  // tile tickets of the self-finishing split-K launch (gemm2_splitk_kernel): SKG_WS_TICKET_BYTES at the end of the registered
  // workspace, zeroed at registration and left zero by every launch (nullptr: the launch is followed by splitk_reduce_kernel)
  unsigned* ws_cnt;
};
constexpr unsigned SKG_FLAG_GN_STATS = 0x8000u;      // internal: set by the launcher when the chosen kernel fuses them

// v2 (gemm2.hip): returns true and launches if the shape is eligible, false otherwise (nothing launched).
bool skg_gemm2_try_launch(const GemmParams& p, int mode, hipStream_t st);
// column width of the tile v2 would use for an M x N output, or 0 if v2 does not take this shape
constexpr size_t SKG_WS_TICKET_BYTES = 16 << 10;      // 4096 tiles (a split-K launch has at most 256)
int skg_gemm2_tile_n(int M, int N, int K, int Cin, int mode, size_t ws_bytes);
// lab build only (tools/lab/gemmws.hip): weight-stationary streaming kernel for N = K = 320 plain GEMMs with M >= 32768
bool skg_gemmws_eligible(const GemmParams& p, int mode);
bool skg_gemmws_try_launch(const GemmParams& p, int mode, hipStream_t st);
// v8 (gemm8.hip): 256 x 320 tiles, one 8-wave workgroup per CU, ping-pong schedule (S1 convolutions of the 64 x 64 level).
bool skg_gemm8_eligible(const GemmParams& p, int mode);
int skg_gemm8_tile_n(const GemmParams& p, int mode);      // 160 / 320, or 0 when v8 does not take the launch
bool skg_gemm8_try_launch(const GemmParams& p, int mode, hipStream_t st);
// lab build only (tools/lab/gemm9.hip, round 5): hand-placed K loop - self-pipelined waves on 32x32x16 MFMAs, fragment register
// double buffer, one barrier per K tile, LDS-DMA spread evenly over the MFMA slots (SKG_GEMM9 selects geometry and schedule variant)
bool skg_gemm9_eligible(const GemmParams& p, int mode);
bool skg_gemm9_try_launch(const GemmParams& p, int mode, hipStream_t st);
// k-pair kernel (gemmk.hip): 128 x 160 tiles, one 8-wave workgroup per CU whose two wave groups take alternate K tiles -
// the launches with at most one tile per CU (DIRECT / S1, plain epilogue, optional split-K across workgroups on top).
bool skg_gemmk_eligible(const GemmParams& p, int mode);
bool skg_gemmk_try_launch(const GemmParams& p, int mode, hipStream_t st);
// out = epi(sum of `splits` fp32 slabs) (gemm2.hip)
void skg_splitk_reduce_launch(const GemmParams& p, const float* ws, int splits, hipStream_t st);
// true when the kernel that skg_gemm8 / skg_gemm2 would run for this launch writes p.gn_partial itself
bool skg_gemm8_fuses_gn(const GemmParams& p, int mode);
bool skg_gemm2_fuses_gn(const GemmParams& p, int mode);
// wino.hip: input / output transforms of the Winograd F(2x2, 3x3) convolution (the GEMM between them is a gemm2.hip split launch)
void skg_wino_in_launch(const half_t* X, int ldx, half_t* V, int rows, int IH, int IW, int Cin, hipStream_t st);
void skg_wino_out_launch(const GemmParams& p, const float* slabs, int IH, int IW, hipStream_t st);
// norms.hip: the stand-alone statistics pass in the same partial format (nch chunks per sample)
void skg_gn_partial_launch(const half_t* X, int ldx, int rows, int HW, int C, int groups, int nch, float* partial,
                           hipStream_t st);

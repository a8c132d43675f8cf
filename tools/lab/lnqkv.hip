// WITHDRAWN EXPERIMENT (round 4; lab build only - `make -C sketch2img_amd/csrc lab`).  Parity-green on its first run (99.85 % of the outputs
// bit-equal to skg_layernorm_fwd + skg_gemm_f16, statistics equal, the pipeline tests green with it) and SLOWER end to end in both forms:
// EXPERIMENTS.md round 4, "norm1 + q / k / v in one launch".  This file holds the second form (2-tile chunks, staged row stores).
//
// norm1 + the fused q / k / v projection of a C = 320 BasicTransformerBlock in ONE row-local launch:
//   Y[m][n] = sum_k LayerNorm(X)[m][k] . W[n][k]        (N = 960: [q | k | v], no bias - diffusers attention.py CrossAttention.to_q/k/v,
//                                                        reached from modules/pipeline.py:96)
// replaces skg_layernorm_fwd (20 us at M = 65 536) + skg_gemm_f16 (72 us, 2.4 TB/s: a 5-step K loop behind a 36 KB operand
// stage) and the [M, 320] tensor between them.  Same formulation as ffblock.hip's proj_out phase: a wave's 16 rows are loaded
// straight into registers, normalised there (norms.hip's two-pass form; the statistics go out for the backward) and ARE the B
// operands of every MFMA; the weights stream through a three-region LDS ring as fragment-major 1 KB pieces (chunk = 2 output
// tiles x 10 k-steps = 20 pieces; every wave issues three operations per chunk, two chunks ahead, counted vmcnt), and after every 20
// output tiles (one of q, k, v) the accumulators leave through the wave's own staging slice as whole 640-byte row segments.
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct LQParams {
  const half_t* X; int ldx;
  half_t* Y; int ldy;
  int M, N;
  const half_t* gamma; const half_t* beta; float eps;
  const half_t* Wp;        // [N / 16][10][512]: piece (u, ks): [lane = 16 g + l][i] = W[16 u + l][32 ks + 8 g + i]
  const half_t* bias;      // [N] or nullptr
  unsigned wbytes;
  float* stats;            // optional [M][2]
};

__global__ __launch_bounds__(512, 1) void ln_gemm_rows_kernel(const LQParams p) {
  constexpr int KS = 10, C = 320, NU = 20, PIECE = 512, CP = 20;      // CP pieces per chunk (2 output tiles)
  constexpr int REG = CP * PIECE;                                      // halves per ring region
  constexpr int DUMP = 3 * REG, STG = DUMP + 4 * PIECE, OP = C + 8;    // [3 regions 60 KB | 4 KB for dead pieces | staging 8 x 16 x OP]
  __shared__ __attribute__((aligned(16))) half_t smem[STG + 8 * 16 * OP];      // 148 KB

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, g = lane >> 4;
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, p.wbytes, 0x00020000);
  const int nchunks = p.N / 32;
  // slot s (0..2) of chunk c: piece q = wave + 8 s -> region c % 3; q >= 20 (the third slot of waves 4..7) or a chunk behind the last one:
  // a dead piece (out of range = zero fill of a slot nobody reads) - every wave issues exactly THREE operations per chunk
  auto dma = [&](int c, int s3) {
    const int q = wave + 8 * s3;
    const bool live = q < CP && c < nchunks;
    const int dst = q < CP ? (c % 3) * REG + q * PIECE : DUMP + (q - CP) * PIECE;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)(smem + dst), 16, live ? (unsigned)lane * 16u : 0x80000000u,
                                             (unsigned)(c * CP + (q < CP ? q : 0)) * 1024u, 0, 0);
  };
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) dma(0, s3);
#pragma unroll
  for (int s3 = 0; s3 < 3; ++s3) dma(1, s3);

  // ---- the wave's 16 rows: load, LayerNorm, keep as B operands
  const int m0 = blockIdx.x * 128 + wave * 16;
  const int mrow = m0 + l16;
  const int mload = min(mrow, p.M - 1);
  half8_t xb[KS];
  {
    const half_t* xr = p.X + (size_t)mload * p.ldx + 8 * g;
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      xb[ks] = ld_half8(xr + 32 * ks);
#pragma unroll
      for (int i = 0; i < 8; ++i) s += (float)xb[ks][i];
    }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.f / C);
    float s2 = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = (float)xb[ks][i] - mean; s2 += d * d; }
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float rstd = rsqrtf(s2 * (1.f / C) + p.eps);
    if (p.stats && g == 0 && mrow < p.M) { p.stats[(size_t)mrow * 2] = mean; p.stats[(size_t)mrow * 2 + 1] = rstd; }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const half8_t gv = ld_half8(p.gamma + 32 * ks + 8 * g), bv = ld_half8(p.beta + 32 * ks + 8 * g);
#pragma unroll
      for (int i = 0; i < 8; ++i) xb[ks][i] = (half_t)(((float)xb[ks][i] - mean) * rstd * (float)gv[i] + (float)bv[i]);
    }
  }

  half_t* const stg = smem + STG + wave * (16 * OP);
  float4_t y[NU];
  for (int c = 0; c < nchunks; ++c) {
    const int jj = c % 10;                // chunk within the 20-tile group (one of q, k, v)
    // chunk c landed (every wave's pieces of it): the three youngest vector-memory operations - chunk c + 1's slots - may still be in
    // flight, everything older (chunk c, the previous group's stores) has completed
    if (c == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    lds_barrier();                        // ... and every wave is done with chunk c - 1, whose region takes chunk c + 2
    if (jj == 0) {
      const int n0 = (c / 10) * C;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const half4_t b = p.bias ? ld_half4(p.bias + n0 + 16 * u + 4 * g) : half4_t{0, 0, 0, 0};
        y[u] = float4_t{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
      }
    }
    const half_t* fr = smem + (c % 3) * REG + lane * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const half8_t w0 = ld_half8(fr + ks * PIECE), w1 = ld_half8(fr + (KS + ks) * PIECE);
      // (the accumulator index must be a compile-time value: ten copies behind a wave-uniform test)
#define LQ_MFMA(J)                                                                                 \
      if (jj == J) {                                                                               \
        y[2 * J] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w0, xb[ks], y[2 * J], 0, 0, 0);           \
        y[2 * J + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(w1, xb[ks], y[2 * J + 1], 0, 0, 0);   \
      }
      LQ_MFMA(0) LQ_MFMA(1) LQ_MFMA(2) LQ_MFMA(3) LQ_MFMA(4) LQ_MFMA(5) LQ_MFMA(6) LQ_MFMA(7) LQ_MFMA(8) LQ_MFMA(9)
#undef LQ_MFMA
      if (ks < 3 && jj != 9) dma(c + 2, ks);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (jj == 9) {
      // ---- the group's 320 columns: accumulators -> the wave's own staging slice -> whole 640-byte row segments; chunk c + 2's slots are
      // issued BEHIND the stores so that they stay the three youngest operations
      const int n0 = (c / 10) * C;
#pragma unroll
      for (int u = 0; u < NU; ++u)
        st_half4(stg + l16 * OP + 16 * u + 4 * g, half4_t{(half_t)y[u][0], (half_t)y[u][1], (half_t)y[u][2], (half_t)y[u][3]});
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-private staging: no workgroup barrier
      constexpr int PPR = C / 8;
#pragma unroll
      for (int j = 0; j < 16 * PPR / 64; ++j) {
        const int pi = lane + 64 * j;
        const int row = pi / PPR, pc = pi - row * PPR;
        if (m0 + row < p.M) st_half8(p.Y + (size_t)(m0 + row) * p.ldy + n0 + pc * 8, ld_half8(stg + row * OP + pc * 8));
      }
#pragma unroll
      for (int s3 = 0; s3 < 3; ++s3) dma(c + 2, s3);
    }
  }
}

}  // namespace

// Y [M][N] = LayerNorm(X; gamma, beta, eps) . W^T (+ bias), K == 320, N % 320 == 0; Wpack fp16 [N / 16][10][512] =
// sketch2img_amd.unet.pack_rows_weight(W); stats (optional) fp32 [M][2] = (mean, rstd) per row, as skg_layernorm_fwd writes them.
// Same rounding points as skg_layernorm_fwd + skg_gemm_f16 (fp16 normalised rows, fp32 accumulation, one fp16 rounding).
extern "C" int skg_ln_gemm_f16(const void* X, int ldx, void* Y, int ldy, int M, int N, int K, const void* gamma, const void* beta,
                               float eps, const void* Wpack, const void* bias, float* stats, void* stream) {
  SKG_REQUIRE(X && Y && gamma && beta && Wpack && M > 0 && K == 320 && N > 0 && N % 320 == 0 && X != Y);
  SKG_REQUIRE(ldx % 8 == 0 && ldx >= K && ldy % 8 == 0 && ldy >= N);
  SKG_REQUIRE(skg_aligned(X, 16) && skg_aligned(Y, 16) && skg_aligned(gamma, 16) && skg_aligned(beta, 16) && skg_aligned(Wpack, 16) &&
              skg_aligned(bias, 8));
  LQParams p;
  p.X = (const half_t*)X; p.ldx = ldx; p.Y = (half_t*)Y; p.ldy = ldy; p.M = M; p.N = N;
  p.gamma = (const half_t*)gamma; p.beta = (const half_t*)beta; p.eps = eps;
  p.Wp = (const half_t*)Wpack; p.bias = (const half_t*)bias; p.wbytes = (unsigned)(N / 16) * 10u * 1024u; p.stats = stats;
  hipLaunchKernelGGL(ln_gemm_rows_kernel, dim3(skg_cdiv(M, 128)), dim3(512), 0, (hipStream_t)stream, p);
  SKG_CHECK_LAUNCH("skg_ln_gemm_f16");
  return SKG_OK;
}

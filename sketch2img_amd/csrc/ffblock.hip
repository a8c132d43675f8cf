// Row-local fused feed-forward sub-block of a BasicTransformerBlock at C = 320 (the 64 x 64 level of SD1.5, the 96 x 96
// level of SD2.1):   Y = X + b2 + W2 . geglu(W1 . LayerNorm(X) + b1)
// replaces four launches - norm3, ff.net.0.proj with the fused gate, ff.net.2 + residual ([diffusers] attention.py
// BasicTransformerBlock.forward, reached from modules/pipeline.py:96 of the reference) - and the two round trips of the
// [M, 1280] gated tensor (168 MB written + read at M = 65 536) between them.
//
// Everything is written TRANSPOSED so that a wave's 16 rows never leave its registers:
//   * a wave owns 16 rows.  Lane (l16, g) loads X[row l16][32 ks + 8 g .. + 7], ks = 0..9: that is already the B operand
//     layout of v_mfma_f32_16x16x32_f16; LayerNorm is a reduction over the lane's 80 values and the 4 lanes g of a row
//     (two shuffles), its fp16 output stays in those 40 registers for the whole kernel;
//   * hidden units come in chunks of 32:  H^T[64 x 16] = W1[val 32 | gate 32 rows][320] . A^T  (4 tiles x 10 MFMAs, the
//     bias is the initial accumulator), the gate runs on the accumulators, and because an accumulator lane holds hidden
//     units 16 t + 4 g + r of row l16, the gated values ARE a B operand again - element i of the k-step <-> hidden unit
//     16 (i >> 2) + 4 g + (i & 3) - for Y^T[320 x 16] += W2[:, chunk] . G^T (20 MFMAs into 80 accumulator registers that
//     live across all chunks), with the same permutation baked into the W2 pack;
//   * only WEIGHTS move: the host packs W1 / W2 "fragment-major" (every 1 KB piece is one MFMA A operand in lane order:
//     unet.pack_ff_block), a chunk is 60 pieces = 60 KB, LDS holds two chunks; a piece is fetched by ONE
//     buffer_load ... lds (1 KB contiguous in memory and in LDS) and read back by ONE conflict-free ds_read_b128.
//     The 2.4 MB pack stays in every XCD's L2.  Per 128-row workgroup that is 7.8 B of operand traffic per kFLOP through
//     the L2 -> LDS path - what the 256 x 320 tile of gemm8.hip pays - and no activation traffic at all.
// One 8-wave workgroup (128 rows) per CU, one barrier per chunk.
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct FFParams {
  const half_t* X; int ldx;
  half_t* Y; int ldy;
  int M;
  const half_t* gamma; const half_t* beta; float eps;
  const half_t* Wp;        // [nch][60][512]: chunk c = 40 W1 pieces (tile t = val0, val1, gate0, gate1; k-step ks) then 20 W2 pieces
  const float* b1p;        // [nch][4][16]: FF1 bias in the order of the W1 tiles
  const half_t* b2;        // [C]
  int nch;
  unsigned wbytes;
  float* stats;            // optional [M][2]: LayerNorm mean / rstd (what the backward of the norm reads)
};

constexpr int MAXCH = 40;

template <int KS>
__global__ __launch_bounds__(512, 1) void ff_block_kernel(const FFParams p) {
  constexpr int C = 32 * KS, NU = C / 16, NP = 4 * KS + NU, PIECE = 512;
  constexpr int STAGE = 64 * PIECE;         // 64 slots: every wave issues 8 pieces per chunk, slots >= NP take zero fills
  constexpr int OP = C + 8;                 // pitch of the epilogue staging rows (halves; 16-byte aligned, 2-way on the 8-byte writes)
  static_assert(KS >= 8 && NP <= 64, "one DMA piece per wave and k-step covers the chunk");
  static_assert(8 * 16 * OP <= 2 * STAGE, "epilogue staging fits the ring");
  __shared__ __attribute__((aligned(16))) half_t smem[2 * STAGE + MAXCH * 64 * 2];    // ONE object (LDS-DMA + ds_read: see gemm2.hip)
  float* const bs = reinterpret_cast<float*>(smem + 2 * STAGE);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, g = lane >> 4;
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, p.wbytes, 0x00020000);
  // piece wave + 8 j of chunk c -> the same slot of `stage`.  Unconditional (a branch per piece would cut the MFMA
  // stream into basic blocks): a dead piece - slot >= NP, or no next chunk - reads out of range = zero fill of its slot
  auto dma_piece = [&](int c, int stage, int j, bool live) {
    const int pc = wave + 8 * j;
    const unsigned voff = (live && pc < NP) ? (unsigned)lane * 16u : 0x80000000u;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)(smem + stage * STAGE + pc * PIECE), 16, voff,
                                             (unsigned)(c * NP + pc) * 1024u, 0, 0);
  };
#pragma unroll
  for (int j = 0; j < 8; ++j) dma_piece(0, 0, j, true);
  for (int i = tid; i < p.nch * 64; i += 512) bs[i] = p.b1p[i];

  // ---- the wave's 16 rows: load, LayerNorm (two passes over registers, as norms.hip), keep as B operands
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
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const half8_t gv = ld_half8(p.gamma + 32 * ks + 8 * g), bv = ld_half8(p.beta + 32 * ks + 8 * g);
#pragma unroll
      for (int i = 0; i < 8; ++i) xb[ks][i] = (half_t)(((float)xb[ks][i] - mean) * rstd * (float)gv[i] + (float)bv[i]);
    }
    if (p.stats && g == 0 && mrow < p.M) { p.stats[(size_t)mrow * 2] = mean; p.stats[(size_t)mrow * 2 + 1] = rstd; }
  }

  float4_t y[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const half4_t b = ld_half4(p.b2 + 16 * u + 4 * g);
    y[u] = float4_t{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
  }

  // ---- one chunk of 32 hidden units out of stage `st`; the next chunk's pieces are fetched under the first product
  auto chunk = [&](const half_t* st, int c, int nstage, bool more) {
    const half_t* fr = st + lane * 8;
    float4_t h[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) h[t] = *reinterpret_cast<const float4_t*>(bs + c * 64 + t * 16 + 4 * g);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
      for (int t = 0; t < 4; ++t)
        h[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld_half8(fr + (t * KS + ks) * PIECE), xb[ks], h[t], 0, 0, 0);
      if (ks < 8) dma_piece(c + 1, nstage, ks, more);
    }
    // gate: f = fp16(W1 a + b1) (the rounding of the unfused FF1 output), out = fp16(f_val * gelu(f_gate)) as gemm2.hip's
    // fused epilogue; element 4 t + r of the B operand <-> hidden unit 16 t + 4 g + r of the chunk
    half8_t gb;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float a = (float)(half_t)h[t][r], gt = (float)(half_t)h[2 + t][r];
        gb[4 * t + r] = (half_t)(a * gelu_fast_f(gt));
      }
#pragma unroll
    for (int u = 0; u < NU; ++u)
      y[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld_half8(fr + (4 * KS + u) * PIECE), gb, y[u], 0, 0, 0);
  };

  for (int c = 0; c < p.nch; c += 2) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    chunk(smem, c, 1, c + 1 < p.nch);
    if (c + 1 < p.nch) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      lds_barrier();
      chunk(smem + STAGE, c + 1, 0, c + 2 < p.nch);
    }
  }

  // ---- epilogue: the wave's Y^T tile through its own slice of the (now idle) ring, then whole-row 16-byte pieces:
  // residual read + store are 10 KB contiguous per wave when the rows are dense
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the last chunk's zero fills must not land in the staging rows
  lds_barrier();
  half_t* const stg = smem + wave * (16 * OP);
  {   // the residual is added in fp32 BEFORE staging (lane-local 8-byte reads of X, L2 hits): one fp16 rounding
    const half_t* xr = p.X + (size_t)mload * p.ldx + 4 * g;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const half4_t r4 = ld_half4(xr + 16 * u);
      const half4_t v = {(half_t)(y[u][0] + (float)r4[0]), (half_t)(y[u][1] + (float)r4[1]), (half_t)(y[u][2] + (float)r4[2]),
                         (half_t)(y[u][3] + (float)r4[3])};
      st_half4(stg + l16 * OP + 16 * u + 4 * g, v);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-private staging: no workgroup barrier
  constexpr int PPR = C / 8;                              // 16-byte pieces per row
#pragma unroll
  for (int j = 0; j < 16 * PPR / 64; ++j) {
    const int pi = lane + 64 * j;
    const int row = pi / PPR, pc = pi - row * PPR;
    if (m0 + row < p.M) st_half8(p.Y + (size_t)(m0 + row) * p.ldy + pc * 8, ld_half8(stg + row * OP + pc * 8));
  }
}

}  // namespace

extern "C" int skg_ff_block_f16(const void* X, int ldx, void* Y, int ldy, int M, int C, int F, const void* gamma,
                                const void* beta, float eps, const void* Wpack, const float* bias1_pack,
                                const void* bias2, float* stats, void* stream) {
  SKG_REQUIRE(X && Y && gamma && beta && Wpack && bias1_pack && bias2 && M > 0);
  SKG_REQUIRE(C == 320 && F % 32 == 0 && F / 32 <= MAXCH && F >= 64);
  SKG_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C);
  SKG_REQUIRE(skg_aligned(X, 16) && skg_aligned(Y, 16) && skg_aligned(gamma, 16) && skg_aligned(beta, 16) &&
              skg_aligned(Wpack, 16) && skg_aligned(bias1_pack, 16) && skg_aligned(bias2, 8));
  FFParams p;
  p.X = (const half_t*)X; p.ldx = ldx; p.Y = (half_t*)Y; p.ldy = ldy; p.M = M;
  p.gamma = (const half_t*)gamma; p.beta = (const half_t*)beta; p.eps = eps;
  p.Wp = (const half_t*)Wpack; p.b1p = bias1_pack; p.b2 = (const half_t*)bias2;
  p.nch = F / 32;
  p.wbytes = (unsigned)p.nch * 60u * 1024u;
  p.stats = stats;
  hipLaunchKernelGGL((ff_block_kernel<10>), dim3(skg_cdiv(M, 128)), dim3(512), 0, (hipStream_t)stream, p);
  SKG_CHECK_LAUNCH("skg_ff_block_f16");
  return SKG_OK;
}

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
// One 8-wave workgroup (128 rows) per CU, one barrier per chunk; the chunks are software-pipelined (the gate of chunk c
// under the first product of chunk c + 1: with everything in lockstep behind one barrier the 8 waves were otherwise all
// in their MFMA phase, then all in their VALU phase - 216 us at M = 65 536 in that form).
#include "common.h"
#include <stdlib.h>

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct FFParams {
  const half_t* X; int ldx;
  half_t* Y; int ldy;
  const half_t* Xl; half_t* Yl;      // accuracy mode (HILO): input and output are PAIRS X + Xl, Y + Yl of fp16 tensors (pitches ldx / ldy)
  int M;
  const half_t* gamma; const half_t* beta; float eps;
  const half_t* Wp;        // [nch][60][512]: chunk c = 40 W1 pieces (tile t = val0, val1, gate0, gate1; k-step ks) then 20 W2 pieces
  const float* b1p;        // [nch][4][16]: FF1 bias in the order of the W1 tiles
  const half_t* b2;        // [C]
  int nch;
  unsigned wbytes;
  float* stats;            // optional [M][2]: LayerNorm mean / rstd (what the backward of the norm reads)
  half_t* keep; int ldkeep; int keep_from;     // optional: rows >= keep_from also store the FF1 output fp16(W1 a + b1), interleaved pack
  int nt_out;              // store Y non-temporally (the output stream then does not evict the weight chunks from L2)
  // PROJ (skg_ff_block_proj_f16): Transformer2DModel.proj_out + the outer residual in the same launch - five more weight chunks
  // behind the nch feed-forward ones (their 40 W1-slot pieces = 4 output tiles x 10 k-steps of W_proj, k order permuted to the
  // accumulator layout), Y = R + bp + W_proj . fp16(X + FF(X)); optionally the GroupNorm partial sums of Y
  const half_t* bp; const half_t* R; int ldr;
  const half_t* Rl;        // HILO + PROJ (skg_ff_block_proj_f16_hilo): the outer residual is the pair R + Rl (pitch ldr)
  float* gn_partial; int gn_hw, gn_groups;
  int nch_w1;              // W1-slot chunks the weight pack holds: nch, or nch + 5 with PROJ
};

constexpr int MAXCH = 40;

// PROBE (timing experiments only, results are wrong; SKG_FFB_PROBE): 1 = no weight DMA inside the loop, 2 = no gate arithmetic,
// 3 = no LDS fragment reads (a register stands in for every A operand), 4 = 1 + 3
// SCHED (SKG_FFB_SCHED, A/B of issue orders): 0 = a gated PAIR after the MFMAs of every second k-step, 1 = the same with the
// two waves of a SIMD in alternate k-steps, 2 = the pair's arithmetic in four stages spread over the MFMAs of two k-steps
// HILO (accuracy mode, skg_ff_block_f16_hilo): LayerNorm reads hi + lo, the residual sum is formed in fp32 on the pair and
// stored as hi = fp16(v), lo = fp16(v - hi); everything between is the same kernel
template <int KS, int PROBE = 0, int SCHED = 0, bool HILO = false, bool PROJ = false>
__global__ __launch_bounds__(512, 1) void ff_block_kernel(const FFParams p) {
  static_assert(!PROJ || (PROBE == 0 && KS == 10), "the proj_out phase exists for the C = 320 kernel");
  constexpr int C = 32 * KS, NU = C / 16, N1 = 4 * KS, NP = N1 + NU, PIECE = 512;
  constexpr int W1ST = N1 * PIECE, W2ST = NU * PIECE;      // halves per ring stage
  constexpr int W2OFF = 2 * W1ST, DUMP = W2OFF + 2 * W2ST, RING = DUMP + (64 - NP) * PIECE;
  constexpr int OP = C + 8;                 // pitch of the epilogue staging rows (halves; 16-byte aligned, 2-way on the 8-byte writes)
  static_assert(KS >= 8 && NP <= 64, "one DMA piece per wave and k-step covers an iteration");
  static_assert(8 * 16 * OP <= RING, "epilogue staging fits the ring");
  // ONE object (LDS-DMA + ds_read: see gemm2.hip): [W1 ring 2 x 40 KB | W2 ring 2 x 20 KB | 4 KB for dead pieces | FF1 bias]
  __shared__ __attribute__((aligned(16))) half_t smem[RING + MAXCH * 64 * 2];
  float* const bs = reinterpret_cast<float*>(smem + RING);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, g = lane >> 4;
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, p.wbytes, 0x00020000);
  // Software pipeline over chunks.  Iteration c computes H(c+1) (W1 ring stage (c+1) & 1) WHILE the gate of chunk c runs on
  // the VALU, then Y += W2(c) . G(c) (W2 ring stage c & 1); meanwhile W1(c+2) lands in W1 stage c & 1 (H(c) was its last
  // reader, one iteration ago) and W2(c+1) in W2 stage (c+1) & 1.  Slot q = wave + 8 j of an iteration: q < 40 a W1 piece,
  // q < 60 a W2 piece, else dead.  Unconditional (a branch per piece would cut the MFMA stream into basic blocks): a dead
  // piece - no such chunk, or q >= 60 - reads out of range = zero fill of a slot nobody reads.
  auto dma_slot = [&](int c, int j) {
    if constexpr (PROBE == 1 || PROBE == 4) return;
    const int q = wave + 8 * j;
    const int cs = q < N1 ? c + 2 : c + 1;                                   // source chunk
    const bool live = q < NP && cs < (PROJ && q < N1 ? p.nch_w1 : p.nch);      // (PROJ: W1(nch) is the first proj_out chunk)
    const int dst = q < N1 ? (c & 1) * W1ST + q * PIECE
                           : (q < NP ? W2OFF + ((c + 1) & 1) * W2ST + (q - N1) * PIECE : DUMP + (q - NP) * PIECE);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)(smem + dst), 16, live ? (unsigned)lane * 16u : 0x80000000u,
                                             (unsigned)(cs * NP + q) * 1024u, 0, 0);
  };
  // before the loop: W1(0) -> W1 stage 0 (iteration -2 of the slot arithmetic; its W2 slots are dead: chunk -1)
#pragma unroll
  for (int j = 0; j < 5; ++j)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)(smem + (wave + 8 * j) * PIECE), 16, (unsigned)lane * 16u,
                                             (unsigned)(wave + 8 * j) * 1024u, 0, 0);
  for (int i = tid; i < p.nch * 64; i += 512) bs[i] = p.b1p[i];

  // ---- the wave's 16 rows: load, LayerNorm (two passes over registers, as norms.hip), keep as B operands
  const int m0 = blockIdx.x * 128 + wave * 16;
  const int mrow = m0 + l16;
  const int mload = min(mrow, p.M - 1);
  half8_t xb[KS];
  if constexpr (HILO)
  {
    // the lo parts are NOT kept in registers across the three passes (40 more live registers spill into the main loop):
    // each pass re-reads them through a pointer the compiler cannot see through (L1 / L2 hits)
    const half_t* xr = p.X + (size_t)mload * p.ldx + 8 * g;
    const half_t* xlr = p.Xl + (size_t)mload * p.ldx + 8 * g;
    float s = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      xb[ks] = ld_half8(xr + 32 * ks);
      const half8_t xl = ld_half8(xlr + 32 * ks);
#pragma unroll
      for (int i = 0; i < 8; ++i) s += (float)xb[ks][i] + (float)xl[i];
    }
    s += __shfl_xor(s, 16, 64);
    s += __shfl_xor(s, 32, 64);
    const float mean = s * (1.f / C);
    asm volatile("" : "+v"(xlr));
    float s2 = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const half8_t xl = ld_half8(xlr + 32 * ks);
#pragma unroll
      for (int i = 0; i < 8; ++i) { const float d = (float)xb[ks][i] + (float)xl[i] - mean; s2 += d * d; }
    }
    s2 += __shfl_xor(s2, 16, 64);
    s2 += __shfl_xor(s2, 32, 64);
    const float rstd = rsqrtf(s2 * (1.f / C) + p.eps);
    asm volatile("" : "+v"(xlr));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const half8_t gv = ld_half8(p.gamma + 32 * ks + 8 * g), bv = ld_half8(p.beta + 32 * ks + 8 * g);
      const half8_t xl = ld_half8(xlr + 32 * ks);
#pragma unroll
      for (int i = 0; i < 8; ++i) xb[ks][i] = (half_t)(((float)xb[ks][i] + (float)xl[i] - mean) * rstd * (float)gv[i] + (float)bv[i]);
    }
    if (p.stats && g == 0 && mrow < p.M) { p.stats[(size_t)mrow * 2] = mean; p.stats[(size_t)mrow * 2 + 1] = rstd; }
  }
  else
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

  // H(cn) = W1(cn) . A^T + b1 out of W1 stage cn & 1 (the bias is the initial accumulator), the DMA slots of iteration c
  // and - GATE - the gate of the previous chunk's accumulators `hin` woven in by hand: after the four MFMAs of k-step ks
  // (independent accumulators, 64 cycles of matrix-pipe time) every second k-step the wave computes two of its eight gated values (~37 VALU
  // instructions) and issues one DMA piece, with the next k-step's fragments already on their way.  The order is pinned
  // with sched_barrier: hipcc alone puts the whole gate in front of (or behind) the MFMAs, and the 8 waves - released
  // together by the barrier - are then all on the VALU, then all on the matrix pipe (sched_group_barrier plans were
  // not honoured for this block).
  // gate: f = fp16(W1 a + b1) (the rounding of the unfused FF1 output), out = fp16(f_val * gelu(f_gate)) as gemm2.hip's
  // fused epilogue; element 4 t + r of the B operand <-> hidden unit 16 t + 4 g + r of the chunk
  // Two adjacent elements per call, so that every conversion is a packed one (v_cvt_pk_f16_f32 rounds two values, the
  // result pair IS one dword of the B operand: no v_and / v_or assembly): ~37 VALU instructions per pair.
  // (The empty asm statements pin the pair's arithmetic between them: without them the IR optimiser merges all the
  // independent gate computations into one block before the machine scheduler ever sees the barriers.)
  auto gate2 = [&](const float4_t (&h)[4], int pr, half8_t& gb) {
    const int t = pr >> 1, r = 2 * (pr & 1);
    float2_t hv = {h[t][r], h[t][r + 1]}, hg = {h[2 + t][r], h[2 + t][r + 1]};
    asm volatile("" : "+v"(hv), "+v"(hg));
    const half2_t pv = __builtin_convertvector(hv, half2_t), pg = __builtin_convertvector(hg, half2_t);
    float2_t o = {(float)pv[0] * gelu_fast_f((float)pg[0]), (float)pv[1] * gelu_fast_f((float)pg[1])};
    if constexpr (PROBE == 2) o = hv + hg;
    half2_t ph = __builtin_convertvector(o, half2_t);
    asm volatile("" : "+v"(ph));
    gb[2 * pr] = ph[0];
    gb[2 * pr + 1] = ph[1];
  };
  // SCHED 2: gelu_fast_f of a pair in four stages (same expressions), the live values pinned at every stage end
  float2_t sa, sg, sz, st, se, sp;
  auto stage = [&](const float4_t (&h)[4], int idx, half8_t& gb) {
    const int pr = idx >> 2;
    switch (idx & 3) {
      case 0: {
        const int t = pr >> 1, r = 2 * (pr & 1);
        float2_t hv = {h[t][r], h[t][r + 1]}, hg = {h[2 + t][r], h[2 + t][r + 1]};
        asm volatile("" : "+v"(hv), "+v"(hg));
        const half2_t pv = __builtin_convertvector(hv, half2_t), pg = __builtin_convertvector(hg, half2_t);
        sa = float2_t{(float)pv[0], (float)pv[1]};
        sg = float2_t{(float)pg[0], (float)pg[1]};
        sz = float2_t{fabsf(sg[0]) * 0.70710678118654752f, fabsf(sg[1]) * 0.70710678118654752f};
        asm volatile("" : "+v"(sa), "+v"(sg), "+v"(sz));
        break;
      }
      case 1: {
        st = float2_t{__builtin_amdgcn_rcpf(fmaf(0.3275911f, sz[0], 1.f)), __builtin_amdgcn_rcpf(fmaf(0.3275911f, sz[1], 1.f))};
        se = float2_t{__builtin_amdgcn_exp2f(-sz[0] * sz[0] * 1.4426950408889634f), __builtin_amdgcn_exp2f(-sz[1] * sz[1] * 1.4426950408889634f)};
        asm volatile("" : "+v"(st), "+v"(se));
        break;
      }
      case 2: {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          float q = fmaf(1.061405429f, st[j], -1.453152027f);
          q = fmaf(q, st[j], 1.421413741f);
          q = fmaf(q, st[j], -0.284496736f);
          sp[j] = fmaf(q, st[j], 0.254829592f);
        }
        asm volatile("" : "+v"(sp));
        break;
      }
      default: {
        float2_t o;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const float e = 1.f - sp[j] * st[j] * se[j];
          o[j] = sa[j] * (0.5f * sg[j] + 0.5f * fabsf(sg[j]) * e);
        }
        if constexpr (PROBE == 2) o = sa + sg;
        half2_t ph = __builtin_convertvector(o, half2_t);
        asm volatile("" : "+v"(ph));
        gb[2 * pr] = ph[0];
        gb[2 * pr + 1] = ph[1];
      }
    }
  };
  const int gphase = (wave >> 2) & 1;
  auto ff1 = [&](float4_t (&h)[4], int cn, int c, const float4_t (&hin)[4], half8_t& gb, bool GATE) {
    const half_t* fr = smem + (cn & 1) * W1ST + lane * 8;
    half8_t fa[4], fb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) h[t] = *reinterpret_cast<const float4_t*>(bs + cn * 64 + t * 16 + 4 * g);
#pragma unroll
    for (int t = 0; t < 4; ++t) fa[t] = (PROBE >= 3) ? xb[t] : ld_half8(fr + (t * KS) * PIECE);
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      half8_t (&cur)[4] = (ks & 1) ? fb : fa;
      half8_t (&nxt)[4] = (ks & 1) ? fa : fb;
      if (ks + 1 < KS) {
#pragma unroll
        for (int t = 0; t < 4; ++t) nxt[t] = (PROBE >= 3) ? xb[(t + ks) % KS] : ld_half8(fr + (t * KS + ks + 1) * PIECE);
      }
      if constexpr (SCHED == 2) {
        h[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[0], xb[ks], h[0], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (GATE && ks < 8) stage(hin, 2 * ks, gb);
        __builtin_amdgcn_sched_barrier(0);
        h[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[1], xb[ks], h[1], 0, 0, 0);
        h[2] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[2], xb[ks], h[2], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
        if (GATE && ks < 8) stage(hin, 2 * ks + 1, gb);
        __builtin_amdgcn_sched_barrier(0);
        h[3] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[3], xb[ks], h[3], 0, 0, 0);
      } else {
#pragma unroll
        for (int t = 0; t < 4; ++t) h[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(cur[t], xb[ks], h[t], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (ks < 8) {
        // the two waves that share a SIMD (w and w + 4) gate in alternate k-steps: one is on the VALU while the other
        // feeds the matrix pipe
        if (SCHED == 0 && GATE && (ks & 1)) gate2(hin, ks >> 1, gb);
        if (SCHED == 1 && GATE && (ks & 1) == gphase) gate2(hin, ks >> 1, gb);
        dma_slot(c, ks);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  // (an opaque base register for the W2 ring: its byte offsets lie beyond the 16-bit immediate of ds_read, and hipcc
  // otherwise rebuilds every fragment address with a VALU instruction)
  int w2base = W2OFF + lane * 8;
  asm volatile("" : "+v"(w2base));
  auto ff2 = [&](half8_t gb, int c) {
    const half_t* fr = smem + (w2base + (c & 1) * W2ST);
#pragma unroll
    for (int u = 0; u < NU; ++u)
      y[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16((PROBE >= 3) ? xb[u % KS] : ld_half8(fr + u * PIECE), gb, y[u], 0, 0, 0);
  };
  // rows whose gate will be differentiated (the cond rows of a guided step) also store the pre-activation in the
  // interleaved FF1 pack order skg_geglu_bwd reads: the lane's units j .. j + 3 (j = 32 c + 16 t + 4 g) are the 8
  // consecutive columns 2 j .. 2 j + 7 = [a a g g a a g g] - one 16-byte store per tile; whole wave tiles (keep_from % 16 == 0)
  const bool keepw = p.keep != nullptr && m0 >= p.keep_from && m0 < p.M;           // wave-uniform
  half_t* const keep_row = p.keep ? p.keep + (size_t)(mload - p.keep_from) * p.ldkeep + 8 * g : nullptr;
  auto store_f = [&](const float4_t (&h)[4], int c) {
    if (!keepw || mrow >= p.M) return;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const half8_t v = {(half_t)h[t][0], (half_t)h[t][1], (half_t)h[2 + t][0], (half_t)h[2 + t][1],
                         (half_t)h[t][2], (half_t)h[t][3], (half_t)h[2 + t][2], (half_t)h[2 + t][3]};
      st_half8(keep_row + 64 * c + 32 * t, v);
    }
  };
  auto iteration = [&](const float4_t (&hin)[4], float4_t (&hout)[4], int c) {
    if (keepw) store_f(hin, c);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    half8_t gb;
    ff1(hout, c + 1, c, hin, gb, true);
    ff2(gb, c);
  };
  // PROJ: W1-slot piece q = wave + 8 s (five per wave) of proj chunk j -> ring region `dst` (halves)
  auto dma_proj = [&](int j, int dst, int s5) {
    const int q = wave + 8 * s5;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)(smem + dst + q * PIECE), 16, (unsigned)lane * 16u,
                                             (unsigned)((p.nch + j) * NP + q) * 1024u, 0, 0);
  };
  auto last = [&](const float4_t (&hin)[4], int c) {
    if (keepw) store_f(hin, c);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();
    if constexpr (PROJ) {      // proj chunk 1 -> the W1 stage whose last reader was H(nch - 1), one iteration ago
#pragma unroll
      for (int s5 = 0; s5 < 5; ++s5) dma_proj(1, ((p.nch + 1) & 1) * W1ST, s5);
    }
    half8_t gb;
#pragma unroll
    for (int pr = 0; pr < 4; ++pr) gate2(hin, pr, gb);
    ff2(gb, c);
  };

  float4_t hc[4], hn[4];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  {
    half8_t unused;
    ff1(hc, 0, -1, hc, unused, false);                  // iteration -1: H(0); fetches W1(1) and W2(0)
  }
  int c = 0;
  for (; c + 2 < p.nch; c += 2) {
    iteration(hc, hn, c);
    iteration(hn, hc, c + 1);
  }
  if (c + 1 < p.nch) {
    iteration(hc, hn, c);
    last(hn, c + 1);
  } else {
    last(hc, c);
  }

  if constexpr (PROJ) {
    // ---- proj_out on the block's output without leaving the registers: p3 = fp16(X + FF) (the rounding of the unfused launch's
    // store) becomes the B operand - accumulator tiles 2 ks, 2 ks + 1 are k-slots 8 g + i <-> channel 32 ks + 16 (i >> 2) + 4 g + (i & 3),
    // the order the proj chunks of the pack are written in - and y restarts from the proj bias.  Chunk nch + j = output tiles
    // 4 j .. 4 j + 3.
    // (HILO: the block output is the pair p3 = hi + lo and proj_out takes both - W . hi + W . lo on the SAME weight fragments: the
    // K-doubled GEMM of the unfused accuracy-mode path at the price of 40 more MFMAs per chunk and no second weight read)
    half8_t yb[KS], ybl[HILO ? KS : 1];
    {
      const half_t* xr = p.X + (size_t)mload * p.ldx + 4 * g;
      const half_t* xlr = HILO ? p.Xl + (size_t)mload * p.ldx + 4 * g : nullptr;
#pragma unroll
      for (int u = 0; u < NU; ++u) {
        const half4_t r4 = ld_half4(xr + 16 * u);
        half4_t l4 = {0, 0, 0, 0};
        if constexpr (HILO) l4 = ld_half4(xlr + 16 * u);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = y[u][r] + (float)r4[r] + (HILO ? (float)l4[r] : 0.f);
          const half_t hi = (half_t)v;
          yb[u >> 1][4 * (u & 1) + r] = hi;
          if constexpr (HILO) ybl[u >> 1][4 * (u & 1) + r] = (half_t)(v - (float)hi);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const half4_t b = ld_half4(p.bp + 16 * u + 4 * g);
      y[u] = float4_t{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
    }
    // three regions take the five chunks: A = W1 stage nch & 1 (chunk 0, fetched by iteration nch - 2; chunk 3), B = the other W1 stage
    // (chunk 1, fetched inside last(); chunk 4), and the two idle W2 stages together (chunk 2) - chunk j + 2 is issued during step j,
    // so a step waits for a chunk issued two steps earlier (counted vmcnt: every wave issues exactly five pieces per chunk)
    const int regA = (p.nch & 1) * W1ST, regB = ((p.nch + 1) & 1) * W1ST;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (j == 0 || j == 4) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(5)" ::: "memory");      // chunk j landed; chunk j + 1 (the five youngest) may still be in flight
      lds_barrier();
      const int reg = (j == 2) ? W2OFF : ((j == 0 || j == 3) ? regA : regB);
      const int regn = (j == 0) ? W2OFF : ((j & 1) ? regA : regB);      // where chunk j + 2 goes: chunk 2 -> W2, 3 -> A, 4 -> B
      const half_t* fr = smem + reg + lane * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        half8_t wf[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) wf[t] = ld_half8(fr + (t * KS + ks) * PIECE);
#pragma unroll
        for (int t = 0; t < 4; ++t) y[4 * j + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t], yb[ks], y[4 * j + t], 0, 0, 0);
        if constexpr (HILO) {
#pragma unroll
          for (int t = 0; t < 4; ++t) y[4 * j + t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[t], ybl[ks], y[4 * j + t], 0, 0, 0);
        }
        if (ks < 5 && j < 3) dma_proj(j + 2, regn, ks);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
  }

  // ---- epilogue: the wave's Y^T tile through its own slice of the (now idle) ring, then whole-row 16-byte pieces:
  // residual read + store are 10 KB contiguous per wave when the rows are dense
  lds_barrier();
  // PROJ: GroupNorm partial sums of the 128 x 320 output tile (layout of gemm2.hip's epilogue: [sample][128-row chunk][group][2]): lane
  // j < 32 of every wave sums its group's 10 channels over the wave's 16 staged rows (fp16 values - HILO: of the hi part -, fp32 sums),
  // the eight waves' figures meet in the dead FF1-bias area and wave 0 adds them in wave order - fixed order, no atomics
  auto gn_from_stage = [&](const half_t* stg) {
    const int cpg = C / p.gn_groups;
    float s1 = 0.f, s2 = 0.f;
    if (lane < p.gn_groups) {
      for (int row = 0; row < 16; ++row) {
        if (m0 + row >= p.M) break;
        const half_t* q = stg + row * OP + lane * cpg;
        for (int i = 0; i < cpg; ++i) { const float v = (float)q[i]; s1 += v; s2 += v * v; }
      }
      bs[(wave * 32 + lane) * 2] = s1;
      bs[(wave * 32 + lane) * 2 + 1] = s2;
    }
    lds_barrier();
    if (wave == 0 && lane < p.gn_groups) {
      float t1 = 0.f, t2 = 0.f;
#pragma unroll
      for (int w = 0; w < 8; ++w) { t1 += bs[(w * 32 + lane) * 2]; t2 += bs[(w * 32 + lane) * 2 + 1]; }
      const int mb = blockIdx.x * 128, b = mb / p.gn_hw, chunk = (mb - b * p.gn_hw) >> 7, nchk = p.gn_hw >> 7;
      float* dst = p.gn_partial + (((size_t)b * nchk + chunk) * p.gn_groups + lane) * 2;
      dst[0] = t1; dst[1] = t2;
    }
  };
  if constexpr (HILO) {
  half_t* const stg = smem + wave * (16 * OP);
  constexpr int PPR = C / 8;                              // 16-byte pieces per row
  {   // the pair residual joins the accumulators first (ONE read of X: Y may alias X), then hi and lo are staged and stored in
      // two passes through the same wave-private slice (LDS operations of a wave execute in order)
      // (PROJ: the OUTER residual pair - the transformer's input - joins here; X + FF went into the proj_out operand above)
    const half_t* xr = PROJ ? p.R + (size_t)mload * p.ldr + 4 * g : p.X + (size_t)mload * p.ldx + 4 * g;
    const half_t* xlr = PROJ ? p.Rl + (size_t)mload * p.ldr + 4 * g : p.Xl + (size_t)mload * p.ldx + 4 * g;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const half4_t r4 = ld_half4(xr + 16 * u), l4 = ld_half4(xlr + 16 * u);
      y[u] += float4_t{(float)r4[0], (float)r4[1], (float)r4[2], (float)r4[3]} + float4_t{(float)l4[0], (float)l4[1], (float)l4[2], (float)l4[3]};
    }
  }
#pragma unroll
  for (int part = 0; part < 2; ++part) {
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const float4_t f = y[u];
      half4_t v = {(half_t)f[0], (half_t)f[1], (half_t)f[2], (half_t)f[3]};
      if (part == 1) v = half4_t{(half_t)(f[0] - (float)v[0]), (half_t)(f[1] - (float)v[1]), (half_t)(f[2] - (float)v[2]), (half_t)(f[3] - (float)v[3])};
      st_half4(stg + l16 * OP + 16 * u + 4 * g, v);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // wave-private staging: no workgroup barrier
    half_t* const dst = part == 0 ? p.Y : p.Yl;
#pragma unroll
    for (int j = 0; j < 16 * PPR / 64; ++j) {
      const int pi = lane + 64 * j;
      const int row = pi / PPR, pc = pi - row * PPR;
      if (m0 + row < p.M) st_half8(dst + (size_t)(m0 + row) * p.ldy + pc * 8, ld_half8(stg + row * OP + pc * 8));
    }
    if constexpr (PROJ) {
      if (part == 0 && p.gn_partial) {      // (workgroup-uniform) statistics of the hi part, before the lo pass reuses the slice
        gn_from_stage(stg);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      }
    }
  }
  } else {
  half_t* const stg = smem + wave * (16 * OP);
  {   // the residual is added in fp32 BEFORE staging (lane-local 8-byte reads of X, L2 hits): one fp16 rounding
    // (PROJ: the OUTER residual - the transformer's input - joins here; X + FF went into the proj_out operand above)
    const half_t* xr = PROJ ? p.R + (size_t)mload * p.ldr + 4 * g : p.X + (size_t)mload * p.ldx + 4 * g;
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
    if (m0 + row < p.M) {
      const half8_t v = ld_half8(stg + row * OP + pc * 8);
      half8_t* dst = reinterpret_cast<half8_t*>(p.Y + (size_t)(m0 + row) * p.ldy + pc * 8);
      if (p.nt_out) __builtin_nontemporal_store(v, dst); else *dst = v;
    }
  }
  if constexpr (PROJ) {
    if (p.gn_partial) gn_from_stage(stg);
  }
  }
}

}  // namespace

static int ff_block_impl(const void* X, const void* Xl, int ldx, void* Y, void* Yl, int ldy, int M, int C, int F, const void* gamma,
                         const void* beta, float eps, const void* Wpack, const float* bias1_pack, const void* bias2,
                         float* stats, void* H, int ldh, int keep_from, void* stream, const void* bias_proj = nullptr,
                         const void* R = nullptr, int ldr = 0, float* gn_partial = nullptr, int gn_hw = 0, int gn_groups = 0,
                         const void* Rl = nullptr) {
  SKG_REQUIRE(X && Y && gamma && beta && Wpack && bias1_pack && bias2 && M > 0 && (Xl != nullptr) == (Yl != nullptr));
  SKG_REQUIRE(!bias_proj || (R && (Rl != nullptr) == (Xl != nullptr) && ldr % 4 == 0 && ldr >= C && skg_aligned(bias_proj, 8) && skg_aligned(R, 8) &&
                             skg_aligned(Rl, 8) && X != Y));
  SKG_REQUIRE(!gn_partial || (bias_proj && gn_groups > 0 && gn_groups <= 32 && C % gn_groups == 0 && gn_hw % 128 == 0 && M % gn_hw == 0));
  SKG_REQUIRE(!H || (ldh % 8 == 0 && ldh >= 2 * F && keep_from >= 0 && keep_from % 16 == 0 && keep_from < M && skg_aligned(H, 16)));
  SKG_REQUIRE(C == 320 && F % 32 == 0 && F / 32 <= MAXCH && F >= 64);
  SKG_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C);
  SKG_REQUIRE(skg_aligned(X, 16) && skg_aligned(Y, 16) && skg_aligned(Xl, 16) && skg_aligned(Yl, 16) && skg_aligned(gamma, 16) &&
              skg_aligned(beta, 16) && skg_aligned(Wpack, 16) && skg_aligned(bias1_pack, 16) && skg_aligned(bias2, 8));
  FFParams p;
  p.X = (const half_t*)X; p.ldx = ldx; p.Y = (half_t*)Y; p.ldy = ldy; p.M = M;
  p.Xl = (const half_t*)Xl; p.Yl = (half_t*)Yl;
  p.gamma = (const half_t*)gamma; p.beta = (const half_t*)beta; p.eps = eps;
  p.Wp = (const half_t*)Wpack; p.b1p = bias1_pack; p.b2 = (const half_t*)bias2;
  p.nch = F / 32;
  p.nch_w1 = p.nch + (bias_proj ? 5 : 0);
  p.wbytes = (unsigned)p.nch_w1 * 60u * 1024u;
  p.bp = (const half_t*)bias_proj; p.R = (const half_t*)R; p.ldr = ldr; p.Rl = (const half_t*)Rl;
  p.gn_partial = gn_partial; p.gn_hw = gn_hw; p.gn_groups = gn_groups;
  p.stats = stats;
  p.keep = (half_t*)H; p.ldkeep = ldh; p.keep_from = keep_from;
  // Y stored non-temporally: alone the kernel is 0.9 % slower (212.3 against 210.5 us), the batch 0.27 % faster (6.723 / 6.725
  // against 6.706 / 6.706 images/s, one box, alternating): the 42 MB output no longer evicts what the next launches read
  p.nt_out = getenv("SKG_FFB_NT") ? atoi(getenv("SKG_FFB_NT")) : 1;      // A/B switch, read per launch
  const dim3 grid(skg_cdiv(M, 128));
#ifdef SKG_LAB      // lab build only (make lab): probe instantiations compute WRONG results (timing experiments, EXPERIMENTS.md round 3)
  static const int probe = getenv("SKG_FFB_PROBE") ? atoi(getenv("SKG_FFB_PROBE")) : 0;
  static const int sched = getenv("SKG_FFB_SCHED") ? atoi(getenv("SKG_FFB_SCHED")) : 0;
  if (!Xl && probe * 10 + sched) {
    switch (probe * 10 + sched) {
      case 10: hipLaunchKernelGGL((ff_block_kernel<10, 1>), grid, dim3(512), 0, (hipStream_t)stream, p); break;
      case 20: hipLaunchKernelGGL((ff_block_kernel<10, 2>), grid, dim3(512), 0, (hipStream_t)stream, p); break;
      case 30: hipLaunchKernelGGL((ff_block_kernel<10, 3>), grid, dim3(512), 0, (hipStream_t)stream, p); break;
      case 40: hipLaunchKernelGGL((ff_block_kernel<10, 4>), grid, dim3(512), 0, (hipStream_t)stream, p); break;
      case 1: hipLaunchKernelGGL((ff_block_kernel<10, 0, 1>), grid, dim3(512), 0, (hipStream_t)stream, p); break;
      default: hipLaunchKernelGGL((ff_block_kernel<10, 0, 2>), grid, dim3(512), 0, (hipStream_t)stream, p); break;
    }
    SKG_CHECK_LAUNCH("skg_ff_block_f16 (probe)");
    return SKG_OK;
  }
#endif
  if (bias_proj && Xl) hipLaunchKernelGGL((ff_block_kernel<10, 0, 0, true, true>), grid, dim3(512), 0, (hipStream_t)stream, p);
  else if (bias_proj) hipLaunchKernelGGL((ff_block_kernel<10, 0, 0, false, true>), grid, dim3(512), 0, (hipStream_t)stream, p);
  else if (Xl) hipLaunchKernelGGL((ff_block_kernel<10, 0, 0, true>), grid, dim3(512), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL((ff_block_kernel<10>), grid, dim3(512), 0, (hipStream_t)stream, p);
  SKG_CHECK_LAUNCH("skg_ff_block_f16");
  return SKG_OK;
}

extern "C" int skg_ff_block_f16_keep(const void* X, int ldx, void* Y, int ldy, int M, int C, int F, const void* gamma,
                                     const void* beta, float eps, const void* Wpack, const float* bias1_pack,
                                     const void* bias2, float* stats, void* H, int ldh, int keep_from, void* stream) {
  return ff_block_impl(X, nullptr, ldx, Y, nullptr, ldy, M, C, F, gamma, beta, eps, Wpack, bias1_pack, bias2, stats, H, ldh, keep_from, stream);
}

extern "C" int skg_ff_block_f16(const void* X, int ldx, void* Y, int ldy, int M, int C, int F, const void* gamma,
                                const void* beta, float eps, const void* Wpack, const float* bias1_pack,
                                const void* bias2, float* stats, void* stream) {
  return ff_block_impl(X, nullptr, ldx, Y, nullptr, ldy, M, C, F, gamma, beta, eps, Wpack, bias1_pack, bias2, stats, nullptr, 0, 0, stream);
}

// ... followed by Transformer2DModel.proj_out + the outer residual in the same launch:
//   Y = R + bias_proj + W_proj . fp16(X + FF(LayerNorm(X)))        (Y must not alias X; it may alias R)
// Wpack holds five more chunks behind the F / 32 feed-forward ones (unet.pack_ff_block(..., w_proj)); stats / H / keep_from as _keep
// (H == NULL: no stash); gn_partial != NULL: also the GroupNorm partial sums of Y, [M / HW][HW / 128][groups][2] as skg_gemm_f16_gn
extern "C" int skg_ff_block_proj_f16(const void* X, int ldx, void* Y, int ldy, int M, int C, int F, const void* gamma, const void* beta,
                                     float eps, const void* Wpack, const float* bias1_pack, const void* bias2, const void* bias_proj,
                                     const void* R, int ldr, float* stats, void* H, int ldh, int keep_from, float* gn_partial, int HW,
                                     int groups, void* stream) {
  SKG_REQUIRE(bias_proj && R);
  return ff_block_impl(X, nullptr, ldx, Y, nullptr, ldy, M, C, F, gamma, beta, eps, Wpack, bias1_pack, bias2, stats, H, ldh, keep_from, stream,
                       bias_proj, R, ldr, gn_partial, HW, groups);
}

// accuracy mode, round 5: skg_ff_block_proj_f16 on pairs - X + X_lo in, Y + Y_lo out, outer residual R + R_lo (pitch ldr); proj_out takes
// the block output as the pair it is (W . hi + W . lo on the same weight fragments); gn_partial: statistics of Y's hi part
extern "C" int skg_ff_block_proj_f16_hilo(const void* X, const void* X_lo, int ldx, void* Y, void* Y_lo, int ldy, int M, int C, int F,
                                          const void* gamma, const void* beta, float eps, const void* Wpack, const float* bias1_pack,
                                          const void* bias2, const void* bias_proj, const void* R, const void* R_lo, int ldr, float* stats,
                                          void* H, int ldh, int keep_from, float* gn_partial, int HW, int groups, void* stream) {
  SKG_REQUIRE(X_lo && Y_lo && bias_proj && R && R_lo);
  return ff_block_impl(X, X_lo, ldx, Y, Y_lo, ldy, M, C, F, gamma, beta, eps, Wpack, bias1_pack, bias2, stats, H, ldh, keep_from, stream,
                       bias_proj, R, ldr, gn_partial, HW, groups, R_lo);
}

// accuracy mode: the same launch on a pair input X + X_lo (pitch ldx) with a pair output Y + Y_lo (pitch ldy); H / keep_from as _keep
extern "C" int skg_ff_block_f16_hilo(const void* X, const void* X_lo, int ldx, void* Y, void* Y_lo, int ldy, int M, int C, int F,
                                     const void* gamma, const void* beta, float eps, const void* Wpack, const float* bias1_pack,
                                     const void* bias2, float* stats, void* H, int ldh, int keep_from, void* stream) {
  SKG_REQUIRE(X_lo && Y_lo);
  return ff_block_impl(X, X_lo, ldx, Y, Y_lo, ldy, M, C, F, gamma, beta, eps, Wpack, bias1_pack, bias2, stats, H, ldh, keep_from, stream);
}

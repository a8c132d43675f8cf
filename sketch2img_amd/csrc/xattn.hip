// Row-local fused cross-attention sub-block of a BasicTransformerBlock at C = 320, 8 heads of 40 (the 64 x 64 level of SD1.5):
//   Y = X + bo + Wo . Attention(Q = Wq . LayerNorm(X), K, V)      with the <= 80 text keys of the row's image
// replaces four launches - norm2, attn2.to_q, the 77-key attention, attn2.to_out + residual ([diffusers] attention.py
// BasicTransformerBlock.forward / CrossAttention, reached from modules/pipeline.py:96; the op order of
// modules/clip_guided_attn.py:127-152) - and the three [M, 320] tensors between them (first half).
//
// Same formulation as ffblock.hip: TRANSPOSED, a wave's 16 rows never leave its registers.  Per head h:
//   Q^T[48 x 16]  = Wq_h[48 x 320] . A^T            3 tiles x 10 MFMAs; rows 40..47 of the head are zero rows of the pack
//   S^T[80 x 16]  = K_h[80 x 48] . Q^T              the accumulator lane (l, g) of Q^T holds d = 16 t + 4 g + r: tiles 0, 1
//                                                   are the B operand of a K = 32 step with k-slot 8 g + i <-> d = 16 (i >> 2) + 4 g + (i & 3),
//                                                   tile 2 is the B operand of a v_mfma_f32_16x16x16_f16 step in natural order
//   softmax over the lane's 20 scores and the 4 lanes of a row (exact maximum, as attn_fwd_short_kernel)
//   O^T[48 x 16]  = V_h^T[48 x 80] . P^T            key tiles (0,1), (2,3) -> two K = 32 steps, tile 4 -> one K = 16 step
//   Y^T[320 x 16] += Wo[:, head h] . O^T            20 output tiles x (K = 32 step on d 0..31 + K = 16 step on d 32..47)
// Every A operand is a "fragment-major" piece of a host-side pack (unet.pack_xattn_weights / pack_xattn_kv): the pack IS the
// LDS image, fetched 1 KB per buffer_load ... lds and read with one conflict-free ds_read_b128 / ds_read_b64 per MFMA.
// One 8-wave workgroup = 128 rows of ONE image (HW % 128 == 0); LDS: Wq_h (30 KB, single: refilled while S / PV / out of the
// head run) + two stages of [K_h 8 KB | V_h 8 KB | Wo_h 30 KB]; two barriers per head.
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef _Float16 half4v __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

struct XAParams {
  const half_t* X; int ldx;
  half_t* Y; int ldy;
  const half_t* Xl; half_t* Yl;      // accuracy mode (HILO): input and output are PAIRS X + Xl, Y + Yl (pitches ldx / ldy)
  int M, HW;
  const half_t* gamma; const half_t* beta; float eps;
  const half_t* Wp;        // [heads][60][512]: Wq_h (30 pieces: tile t, k-step ks) then the Wo_h image (30 pieces)
  const half_t* KVp;       // [images][heads][16][512]: K_h image (8 pieces) then V_h image (8 pieces)
  const half_t* bo;        // [C]
  int heads, nkv;
  float sc;                // dh^-0.5 * log2(e)
  unsigned wbytes, kvbytes;
  // stashing launch (skg_xattn_block_f16_keep): rows >= keep_from (whole images) also store what the backward of the four
  // replaced launches reads - norm2's (mean, rstd), the to_q output, the attention output and the log-sum-exp, indexed from keep_from
  float* kstats; half_t* kq; half_t* ko; int ldk; float* klse; int keep_from;
};

constexpr float XA_NEG = -30000.f;

// A K = 16 step whose operands are 4 halves per lane.  XA_K16 = 1: v_mfma_f32_16x16x16_f16; 0 (default): the K = 32
// instruction on zero-extended operands - slot (g, i < 4) of A meets slot (g, i) of B whatever the instruction's
// internal k order is, the upper slots contribute 0 * 0.
#ifndef XA_K16
#define XA_K16 0
#endif
__device__ __forceinline__ float4_t mfma_k16(half4v a, half4v b, float4_t c) {
#if XA_K16
  return __builtin_amdgcn_mfma_f32_16x16x16f16(a, b, c, 0, 0, 0);
#else
  const half8_t a8 = {a[0], a[1], a[2], a[3], 0, 0, 0, 0}, b8 = {b[0], b[1], b[2], b[3], 0, 0, 0, 0};
  return __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c, 0, 0, 0);
#endif
}

// An MFMA that starts a fresh accumulator: its operands stay live past it (the empty asm uses them), so that hipcc cannot
// allocate the destination on top of a dying A / B fragment.  Found on the device with structured-V probes: a
// v_mfma_f32_16x16x32_f16 whose destination overlapped its A operand, followed closely by the dependent next k-step of the same
// accumulator, lost that next step's contribution in accumulator registers 0, 1 for part of the waves (timing dependent).
// ffblock.hip / attention.hip have such overlaps too, but never a dependent MFMA right behind one.
__device__ __forceinline__ float4_t mfma32_fresh(half8_t a, half8_t b, float4_t c) {
  const float4_t d = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  asm volatile("" ::"v"(a), "v"(b));
  return d;
}

// HILO (accuracy mode, skg_xattn_block_f16_hilo): LayerNorm reads hi + lo, the residual sum is formed in fp32 on the pair and stored
// as hi = fp16(v), lo = fp16(v - hi); everything between is the same kernel
// DH = 64 (round 5: the 5 x 64 heads of SD2.1's C = 320 blocks): no K = 16 tails on the head width - Q^T is 4 tiles, S^T and the
// out-projection take two K = 32 steps over d, O^T is 4 tiles; only the key axis (80 = 2 x 32 + 16) keeps its K = 16 step.  The LDS image is
// 40 Wq pieces + two stages of [K 10 | V 10 | Wo 40] = 160 KB exactly: dead DMA slots are skipped (wave-uniform branch) instead of dumped.
template <bool HILO, bool KEEP = false, int DH = 40>
__global__ __launch_bounds__(512, 1) void xattn_block_kernel(const XAParams p) {
  constexpr int KS = 10, C = 320, NU = 20, PIECE = 512;
  constexpr bool D64 = DH == 64;
  static_assert(DH == 40 || DH == 64, "head width");
  constexpr int NTQ = D64 ? 4 : 3;                       // 16-row tiles of Q^T / O^T
  constexpr int WQ = 0, NWQ = NTQ * KS;                  // pieces
  constexpr int NKP = D64 ? 10 : 8, NVP = D64 ? 10 : 8;  // K / V image pieces per (image, head)
  constexpr int NWO = D64 ? 40 : 30, WPH = NWQ + NWO;    // Wo image pieces; pack pieces per head
  constexpr int STG = NWQ * PIECE, NST = NKP + NVP + NWO;      // stage = [K | V | Wo]: 46 / 60 pieces
  constexpr int KOFF = 0, VOFF = NKP * PIECE, WOOFF = (NKP + NVP) * PIECE;
  constexpr int DUMP = STG + 2 * NST * PIECE;            // (DH = 40) 2 pieces for dead DMA slots
  constexpr int LDSH = DUMP + (D64 ? 0 : 2 * PIECE);
  constexpr int NSTS = (NST + 7) / 8, NWQS = (NWQ + 7) / 8;    // DMA slots per wave: stage 6 / 8, Wq 4 / 5
  constexpr int OP = C + 8;
  static_assert(8 * 16 * OP <= LDSH, "epilogue staging fits");
  __shared__ __attribute__((aligned(16))) half_t smem[LDSH];        // ONE object (LDS-DMA + ds_read: see gemm2.hip)

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, g = lane >> 4;
  const int img = (blockIdx.x * 128) / p.HW;
  const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)p.Wp, 0, p.wbytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rKV = __builtin_amdgcn_make_buffer_rsrc((void*)p.KVp, 0, p.kvbytes, 0x00020000);
  // slot j (0..5) of the stage fetch of head h: piece q = wave + 8 j of [K | V | Wo] -> stage h & 1; dead: q >= 46 or no such head
  auto dma_stage = [&](int h, int j) {
    const int q = wave + 8 * j;
    const bool live = q < NST && h < p.heads;
    if constexpr (D64) {
      if (!live) return;
    }
    const int dst = q < NST ? STG + (h & 1) * NST * PIECE + q * PIECE : DUMP + (q - NST) * PIECE;
    const unsigned voff = live ? (unsigned)lane * 16u : 0x80000000u;
    if (q < NKP + NVP)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rKV, (lds_ptr_t)(smem + dst), 16, voff,
                                               (unsigned)(((img * p.heads + h) * (NKP + NVP) + q) * 1024), 0, 0);
    else
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)(smem + dst), 16, voff, (unsigned)((h * WPH + NWQ + (q - (NKP + NVP))) * 1024), 0, 0);
  };
  // slot j (0..3) of the Wq fetch of head h: piece q = wave + 8 j
  auto dma_wq = [&](int h, int j) {
    const int q = wave + 8 * j;
    const bool live = q < NWQ && h < p.heads;
    if constexpr (D64) {
      if (!live) return;
    }
    const int dst = q < NWQ ? WQ + q * PIECE : DUMP + (q - NWQ) * PIECE;
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)(smem + dst), 16, live ? (unsigned)lane * 16u : 0x80000000u,
                                             (unsigned)((h * WPH + q) * 1024), 0, 0);
  };
#pragma unroll
  for (int j = 0; j < NWQS; ++j) dma_wq(0, j);
#pragma unroll
  for (int j = 0; j < NSTS; ++j) dma_stage(0, j);

  // ---- the wave's 16 rows: load, LayerNorm (as norms.hip), keep as B operands
  const int m0 = blockIdx.x * 128 + wave * 16;
  const int mrow = m0 + l16;
  const int mload = min(mrow, p.M - 1);
  half8_t xb[KS];
  if constexpr (HILO)
  {
    // (the lo parts are re-read in each of the three passes through a pointer the compiler cannot see through: see ffblock.hip)
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
    if constexpr (KEEP) {
      if (m0 >= p.keep_from && g == 0 && mrow < p.M) {
        p.kstats[(size_t)(mrow - p.keep_from) * 2] = mean;
        p.kstats[(size_t)(mrow - p.keep_from) * 2 + 1] = rstd;
      }
    }
    asm volatile("" : "+v"(xlr));
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const half8_t gv = ld_half8(p.gamma + 32 * ks + 8 * g), bv = ld_half8(p.beta + 32 * ks + 8 * g);
      const half8_t xl = ld_half8(xlr + 32 * ks);
#pragma unroll
      for (int i = 0; i < 8; ++i) xb[ks][i] = (half_t)(((float)xb[ks][i] + (float)xl[i] - mean) * rstd * (float)gv[i] + (float)bv[i]);
    }
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
    if constexpr (KEEP) {
      if (m0 >= p.keep_from && g == 0 && mrow < p.M) {
        p.kstats[(size_t)(mrow - p.keep_from) * 2] = mean;
        p.kstats[(size_t)(mrow - p.keep_from) * 2 + 1] = rstd;
      }
    }
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const half8_t gv = ld_half8(p.gamma + 32 * ks + 8 * g), bv = ld_half8(p.beta + 32 * ks + 8 * g);
#pragma unroll
      for (int i = 0; i < 8; ++i) xb[ks][i] = (half_t)(((float)xb[ks][i] - mean) * rstd * (float)gv[i] + (float)bv[i]);
    }
  }
  float4_t y[NU];
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const half4_t b = ld_half4(p.bo + 16 * u + 4 * g);
    y[u] = float4_t{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
  }
  // Fresh accumulators start from an OPAQUE zero register set, never from the inline constant 0: with C = 0 hipcc may give
  // v_mfma_f32_16x16x32_f16 a destination that overlaps its own A operand (the LDS fragment dies there), and on gfx950 the
  // result is then wrong for part of the K range (found with structured-V probes: keys 32..63 lost in accumulator registers
  // 0, 1 of one tile, timing dependent).  ffblock.hip never hits the pattern - its accumulators start from bias registers.
  auto fresh = [&]() {
    float4_t z = {0.f, 0.f, 0.f, 0.f};
    asm volatile("" : "+v"(z));
    return z;
  };

  for (int h = 0; h < p.heads; ++h) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    lds_barrier();                                           // A: Wq_h and stage h landed; every wave is done with head h - 1
    // ---- Q^T = Wq_h . A^T; the next head's [K | V | Wo] is fetched under it (its stage was head h - 1's)
    float4_t q[NTQ];
#pragma unroll
    for (int t = 0; t < NTQ; ++t) q[t] = fresh();
    {
      const half_t* fr = smem + WQ + lane * 8;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int t = 0; t < NTQ; ++t)
          q[t] = ks == 0 ? mfma32_fresh(ld_half8(fr + (t * KS + ks) * PIECE), xb[ks], q[t])
                         : __builtin_amdgcn_mfma_f32_16x16x32_f16(ld_half8(fr + (t * KS + ks) * PIECE), xb[ks], q[t], 0, 0, 0);
        if (ks < NSTS) dma_stage(h + 1, ks);
      }
    }
    lds_barrier();                                           // B: every wave has read Wq_h - its region takes Wq_{h+1}
    // q: fp16 (the rounding of the stored to_q output), then scaled and rounded again (attn_fwd_short_kernel's qf)
    half8_t qb32, qb32b;      // (qb32b: d 32..63 of a 64-wide head)
    half4v qb16;
#pragma unroll
    for (int i = 0; i < 8; ++i) qb32[i] = (half_t)((float)(half_t)q[i >> 2][i & 3] * p.sc);
    if constexpr (D64) {
#pragma unroll
      for (int i = 0; i < 8; ++i) qb32b[i] = (half_t)((float)(half_t)q[2 + (i >> 2)][i & 3] * p.sc);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) qb16[i] = (half_t)((float)(half_t)q[2][i] * p.sc);
    }
    if constexpr (KEEP) {      // the to_q output of head h, fp16 as the replaced GEMM stores it: d = 16 t + 4 g + r of row l16
      if (m0 >= p.keep_from && mrow < p.M) {
        half_t* qr = p.kq + (size_t)(mrow - p.keep_from) * p.ldk + h * DH + 4 * g;
#pragma unroll
        for (int t = 0; t < NTQ; ++t)
          if (D64 || t < 2 || g < 2) st_half4(qr + 16 * t, half4_t{(half_t)q[t][0], (half_t)q[t][1], (half_t)q[t][2], (half_t)q[t][3]});
      }
    }
    const half_t* st = smem + STG + (h & 1) * NST * PIECE;
    // ---- S^T = K_h . Q^T: 5 key tiles
    float4_t s[5];
#pragma unroll
    for (int kt = 0; kt < 5; ++kt) s[kt] = mfma32_fresh(ld_half8(st + KOFF + kt * PIECE + lane * 8), qb32, fresh());
    if constexpr (D64) {
#pragma unroll
      for (int kt = 0; kt < 5; ++kt)
        s[kt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld_half8(st + KOFF + (5 + kt) * PIECE + lane * 8), qb32b, s[kt], 0, 0, 0);
    } else {
#pragma unroll
      for (int kt = 0; kt < 5; ++kt)
        s[kt] = mfma_k16(*reinterpret_cast<const half4v*>(st + KOFF + 5 * PIECE + kt * 256 + lane * 4), qb16, s[kt]);
    }
    dma_wq(h + 1, 0);
    dma_wq(h + 1, 1);
    // keys behind nkv are masked (lane holds keys 16 kt + 4 g + r)
#pragma unroll
    for (int kt = 0; kt < 5; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (16 * kt + 4 * g + r >= p.nkv) s[kt][r] = XA_NEG;
    float mx = s[0][0];
#pragma unroll
    for (int kt = 0; kt < 5; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[kt][r]);
    mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float li = 0.f;
#pragma unroll
    for (int kt = 0; kt < 5; ++kt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[kt][r] - mx);
        s[kt][r] = e;
        li += e;
      }
    li += __shfl_xor(li, 16, 64);
    li += __shfl_xor(li, 32, 64);
    half8_t pb32[2];
    half4v pb16;
#pragma unroll
    for (int sv = 0; sv < 2; ++sv)
#pragma unroll
      for (int i = 0; i < 8; ++i) pb32[sv][i] = (half_t)s[2 * sv + (i >> 2)][i & 3];
#pragma unroll
    for (int i = 0; i < 4; ++i) pb16[i] = (half_t)s[4][i];
    dma_wq(h + 1, 2);
    dma_wq(h + 1, 3);
    if constexpr (NWQS > 4) dma_wq(h + 1, 4);
    // ---- O^T = V_h^T . P^T: 3 / 4 tiles of d
    float4_t o[NTQ];
#pragma unroll
    for (int dt = 0; dt < NTQ; ++dt) o[dt] = mfma32_fresh(ld_half8(st + VOFF + (dt * 2) * PIECE + lane * 8), pb32[0], fresh());
#pragma unroll
    for (int dt = 0; dt < NTQ; ++dt)
      o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld_half8(st + VOFF + (dt * 2 + 1) * PIECE + lane * 8), pb32[1], o[dt], 0, 0, 0);
#pragma unroll
    for (int dt = 0; dt < NTQ; ++dt)
      o[dt] = mfma_k16(*reinterpret_cast<const half4v*>(st + VOFF + 2 * NTQ * PIECE + dt * 256 + lane * 4), pb16, o[dt]);
    const float inv = 1.f / li;
    half8_t ob32, ob32b;
    half4v ob16;
#pragma unroll
    for (int i = 0; i < 8; ++i) ob32[i] = (half_t)(o[i >> 2][i & 3] * inv);
    if constexpr (D64) {
#pragma unroll
      for (int i = 0; i < 8; ++i) ob32b[i] = (half_t)(o[2 + (i >> 2)][i & 3] * inv);
    } else {
#pragma unroll
      for (int i = 0; i < 4; ++i) ob16[i] = (half_t)(o[2][i] * inv);
    }
    if constexpr (KEEP) {      // the attention output (normalised, fp16) and lse = ln sum_k exp(scale q.k) of (row, head)
      if (m0 >= p.keep_from && mrow < p.M) {
        half_t* orow = p.ko + (size_t)(mrow - p.keep_from) * p.ldk + h * DH + 4 * g;
        st_half4(orow, half4_t{ob32[0], ob32[1], ob32[2], ob32[3]});
        st_half4(orow + 16, half4_t{ob32[4], ob32[5], ob32[6], ob32[7]});
        if constexpr (D64) {
          st_half4(orow + 32, half4_t{ob32b[0], ob32b[1], ob32b[2], ob32b[3]});
          st_half4(orow + 48, half4_t{ob32b[4], ob32b[5], ob32b[6], ob32b[7]});
        } else if (g < 2) st_half4(orow + 32, half4_t{ob16[0], ob16[1], ob16[2], ob16[3]});
        if (g == 0) {
          const int mk = mrow - p.keep_from, bk = mk / p.HW;
          p.klse[((size_t)bk * p.heads + h) * p.HW + (mk - bk * p.HW)] = (log2f(li) + mx) * 0.6931471805599453f;
        }
      }
    }
    // ---- Y^T += Wo[:, head h] . O^T
#pragma unroll
    for (int u = 0; u < NU; ++u) y[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld_half8(st + WOOFF + u * PIECE + lane * 8), ob32, y[u], 0, 0, 0);
    if constexpr (D64) {
#pragma unroll
      for (int u = 0; u < NU; ++u)
        y[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ld_half8(st + WOOFF + (NU + u) * PIECE + lane * 8), ob32b, y[u], 0, 0, 0);
    } else {
#pragma unroll
      for (int u = 0; u < NU; ++u)
        y[u] = mfma_k16(*reinterpret_cast<const half4v*>(st + WOOFF + 20 * PIECE + u * 256 + lane * 4), ob16, y[u]);
    }
  }

  // ---- epilogue (as ffblock.hip): residual added in fp32, the tile through the wave's own slice of the idle LDS, whole-row stores
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  lds_barrier();
  if constexpr (HILO) {
    half_t* const stg = smem + wave * (16 * OP);
    constexpr int PPR = C / 8;
    {   // the pair residual joins the accumulators first (ONE read of X: Y may alias X); hi, then lo, through the same slice
      const half_t* xr = p.X + (size_t)mload * p.ldx + 4 * g;
      const half_t* xlr = p.Xl + (size_t)mload * p.ldx + 4 * g;
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
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      half_t* const dst = part == 0 ? p.Y : p.Yl;
#pragma unroll
      for (int j = 0; j < 16 * PPR / 64; ++j) {
        const int pi = lane + 64 * j;
        const int row = pi / PPR, pc = pi - row * PPR;
        if (m0 + row < p.M) st_half8(dst + (size_t)(m0 + row) * p.ldy + pc * 8, ld_half8(stg + row * OP + pc * 8));
      }
    }
  } else {
  half_t* const stg = smem + wave * (16 * OP);
  {
    const half_t* xr = p.X + (size_t)mload * p.ldx + 4 * g;
#pragma unroll
    for (int u = 0; u < NU; ++u) {
      const half4_t r4 = ld_half4(xr + 16 * u);
      const half4_t v = {(half_t)(y[u][0] + (float)r4[0]), (half_t)(y[u][1] + (float)r4[1]), (half_t)(y[u][2] + (float)r4[2]),
                         (half_t)(y[u][3] + (float)r4[3])};
      st_half4(stg + l16 * OP + 16 * u + 4 * g, v);
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  constexpr int PPR = C / 8;
#pragma unroll
  for (int j = 0; j < 16 * PPR / 64; ++j) {
    const int pi = lane + 64 * j;
    const int row = pi / PPR, pc = pi - row * PPR;
    if (m0 + row < p.M) st_half8(p.Y + (size_t)(m0 + row) * p.ldy + pc * 8, ld_half8(stg + row * OP + pc * 8));
  }
  }
}

}  // namespace

static int xattn_block_impl(const void* X, const void* Xl, int ldx, void* Y, void* Yl, int ldy, int M, int HW, int C, int heads,
                            int Nkv, const void* gamma, const void* beta, float eps, const void* Wpack, const void* KVpack,
                            const void* bias_out, float scale, void* stream, float* kstats = nullptr, void* kq = nullptr,
                            void* ko = nullptr, int ldk = 0, float* klse = nullptr, int keep_from = 0) {
  SKG_REQUIRE(X && Y && gamma && beta && Wpack && KVpack && bias_out && M > 0 && (Xl != nullptr) == (Yl != nullptr));
  SKG_REQUIRE(!kq || (kstats && ko && klse && ldk % 4 == 0 && ldk >= C && keep_from >= 0 && keep_from < M && HW > 0 &&
                      keep_from % HW == 0 && skg_aligned(kq, 8) && skg_aligned(ko, 8)));
  SKG_REQUIRE(C == 320 && (heads == 8 || heads == 5) && Nkv > 0 && Nkv <= 80 && HW > 0 && HW % 128 == 0 && M % HW == 0);
  const bool d64 = heads == 5;      // SD2.1: 5 heads of 64 (pack_xattn_weights / pack_xattn_kv lay the images out per head width)
  SKG_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C);
  SKG_REQUIRE(skg_aligned(X, 16) && skg_aligned(Y, 16) && skg_aligned(Xl, 16) && skg_aligned(Yl, 16) && skg_aligned(gamma, 16) &&
              skg_aligned(beta, 16) && skg_aligned(Wpack, 16) && skg_aligned(KVpack, 16) && skg_aligned(bias_out, 8));
  XAParams p;
  p.X = (const half_t*)X; p.ldx = ldx; p.Y = (half_t*)Y; p.ldy = ldy; p.M = M; p.HW = HW;
  p.Xl = (const half_t*)Xl; p.Yl = (half_t*)Yl;
  p.gamma = (const half_t*)gamma; p.beta = (const half_t*)beta; p.eps = eps;
  p.Wp = (const half_t*)Wpack; p.KVp = (const half_t*)KVpack; p.bo = (const half_t*)bias_out;
  p.heads = heads; p.nkv = Nkv;
  p.sc = scale * 1.4426950408889634f;
  p.wbytes = (unsigned)heads * (d64 ? 80u : 60u) * 1024u;
  p.kvbytes = (unsigned)(M / HW) * (unsigned)heads * (d64 ? 20u : 16u) * 1024u;
  p.kstats = kstats; p.kq = (half_t*)kq; p.ko = (half_t*)ko; p.ldk = ldk; p.klse = klse; p.keep_from = keep_from;
  if (d64) {
    if (kq && Xl) hipLaunchKernelGGL((xattn_block_kernel<true, true, 64>), dim3(M / 128), dim3(512), 0, (hipStream_t)stream, p);
    else if (kq) hipLaunchKernelGGL((xattn_block_kernel<false, true, 64>), dim3(M / 128), dim3(512), 0, (hipStream_t)stream, p);
    else if (Xl) hipLaunchKernelGGL((xattn_block_kernel<true, false, 64>), dim3(M / 128), dim3(512), 0, (hipStream_t)stream, p);
    else hipLaunchKernelGGL((xattn_block_kernel<false, false, 64>), dim3(M / 128), dim3(512), 0, (hipStream_t)stream, p);
  } else if (kq && Xl) hipLaunchKernelGGL((xattn_block_kernel<true, true>), dim3(M / 128), dim3(512), 0, (hipStream_t)stream, p);
  else if (kq) hipLaunchKernelGGL((xattn_block_kernel<false, true>), dim3(M / 128), dim3(512), 0, (hipStream_t)stream, p);
  else if (Xl) hipLaunchKernelGGL(xattn_block_kernel<true>, dim3(M / 128), dim3(512), 0, (hipStream_t)stream, p);
  else hipLaunchKernelGGL(xattn_block_kernel<false>, dim3(M / 128), dim3(512), 0, (hipStream_t)stream, p);
  SKG_CHECK_LAUNCH("skg_xattn_block_f16");
  return SKG_OK;
}

extern "C" int skg_xattn_block_f16(const void* X, int ldx, void* Y, int ldy, int M, int HW, int C, int heads, int Nkv,
                                   const void* gamma, const void* beta, float eps, const void* Wpack, const void* KVpack,
                                   const void* bias_out, float scale, void* stream) {
  return xattn_block_impl(X, nullptr, ldx, Y, nullptr, ldy, M, HW, C, heads, Nkv, gamma, beta, eps, Wpack, KVpack, bias_out, scale, stream);
}

// accuracy mode: pair input X + X_lo (pitch ldx), pair output Y + Y_lo (pitch ldy)
extern "C" int skg_xattn_block_f16_hilo(const void* X, const void* X_lo, int ldx, void* Y, void* Y_lo, int ldy, int M, int HW, int C,
                                        int heads, int Nkv, const void* gamma, const void* beta, float eps, const void* Wpack,
                                        const void* KVpack, const void* bias_out, float scale, void* stream) {
  SKG_REQUIRE(X_lo && Y_lo);
  return xattn_block_impl(X, X_lo, ldx, Y, Y_lo, ldy, M, HW, C, heads, Nkv, gamma, beta, eps, Wpack, KVpack, bias_out, scale, stream);
}

// the stashing launch of a guided step: rows >= keep_from (a multiple of HW: whole images, the cond half) also store norm2's
// statistics [M - keep_from][2], the to_q output Q and the attention output O ([M - keep_from][ldk], head h in columns
// 40 h .. 40 h + 39) and lse [(M - keep_from) / HW][heads][HW] (natural log, skg_attn_fwd's convention) - what
// skg_attn_bwd_dq / skg_layernorm_bwd of the four replaced launches read
extern "C" int skg_xattn_block_f16_keep(const void* X, int ldx, void* Y, int ldy, int M, int HW, int C, int heads, int Nkv,
                                        const void* gamma, const void* beta, float eps, const void* Wpack, const void* KVpack,
                                        const void* bias_out, float scale, float* stats, void* Q, void* O, int ldk, float* lse,
                                        int keep_from, void* stream) {
  SKG_REQUIRE(stats && Q && O && lse);
  return xattn_block_impl(X, nullptr, ldx, Y, nullptr, ldy, M, HW, C, heads, Nkv, gamma, beta, eps, Wpack, KVpack, bias_out, scale, stream,
                          stats, Q, O, ldk, lse, keep_from);
}

// ... on pairs (accuracy mode, round 5): skg_xattn_block_f16_hilo that also stashes what the backward of the cond rows reads
extern "C" int skg_xattn_block_f16_hilo_keep(const void* X, const void* X_lo, int ldx, void* Y, void* Y_lo, int ldy, int M, int HW, int C,
                                             int heads, int Nkv, const void* gamma, const void* beta, float eps, const void* Wpack,
                                             const void* KVpack, const void* bias_out, float scale, float* stats, void* Q, void* O, int ldk,
                                             float* lse, int keep_from, void* stream) {
  SKG_REQUIRE(X_lo && Y_lo && stats && Q && O && lse);
  return xattn_block_impl(X, X_lo, ldx, Y, Y_lo, ldy, M, HW, C, heads, Nkv, gamma, beta, eps, Wpack, KVpack, bias_out, scale, stream,
                          stats, Q, O, ldk, lse, keep_from);
}

// Flash-style fused attention for gfx950, fp16 in / fp32 accumulate: forward, dQ and dK/dV.
//
// Formulation ("swapped" products, so every softmax statistic is lane-local or a 2-step shuffle):
// a wave owns 16 query rows and computes S^T = K Q^T with 16x16x32 f16 MFMAs, K rows as the A
// operand (from LDS) and the wave's Q rows as the B operand (registers, loaded once).  In the MFMA
// C layout lane (q = lane & 15, g = lane >> 4) then holds S^T[kv = 16 t + 4 g + r][q] in register r of
// tile t: all 16 scores a lane holds belong to ONE query, so max / sum are in-lane plus two xor
// shuffles (16, 32), and the online-softmax rescale factor is a per-lane scalar.
// For the second product O^T = V^T P^T the probabilities are used straight from those registers as
// the B operand: element i of lane (q, g) is declared to be k-slot (g, i) <-> key
// 32 s + 16 (i >> 2) + 4 g + (i & 3); the MFMA only needs A and B to agree on the k <-> key map, so
// the A operand (V^T rows = head-dim index) is read with the same map: two ds_read_b64_tr_b16 per fragment out of
// the ROW-MAJOR V tile (skg_attn_fwd_rowv; tfrag_rows below), or two 8-byte reads out of a tile stored transposed
// when the caller hands over V^T (skg_attn_fwd, _causal).  No P round trip through LDS, no cross-lane transposes.
// The backward kernels read K^T, Q^T and dO^T fragments the same way from the row tiles they stage anyway, so
// nothing on the UNet path needs skg_transpose_f16 any more.
//
// Block = 4 waves = 64 query rows (forward, dQ) or 64 key rows (dK/dV); KV / Q tiles of 64 rows are
// staged in LDS.  Pitches: row tiles [64][32 KS + 16] (ds_read_b128 is served in four NON-contiguous 16-lane groups,
// MI355X_MICROARCH.md section LDS: a pitch of 16 or 48 mod 64 halves puts each group on 16 distinct 16-byte slots; the
// round-1 pitch 32 KS + 8 was 2-way conflicted - SQ_LDS_BANK_CONFLICT = 42 % of SQ_LDS_IDX_ACTIVE), transposed tiles
// [d][72] read with ds_read_b64 pairs (two 32-lane groups: conflict-free at 72).
#include "common.h"
#include <stdlib.h>
#include <type_traits>

namespace {

struct AttnParams {
  const half_t* Q; int ldq;
  const half_t* K; int ldk;
  const half_t* V; int ldv;       // row-major V (backward)
  const half_t* Vt; int ldvt;     // forward: V^T (skg_attn_fwd, _causal) or the row-major V and its pitch (skg_attn_fwd_rowv)
  const half_t* dO; int lddo;
  half_t* O; int ldo;             // fwd: O; dq: dQ; dkv: dK
  half_t* O2; int ldo2;           // dkv: dV
  float* lse;                     // [batch][heads][Nq]
  const float* delta;             // [batch][heads][Nq]
  const half_t* Of; int ldof;     // dq with the delta prologue (skg_attn_bwd_dq_delta): the forward's O, and where delta is also stored
  float* delta_out;
  int batch, heads, Nq, Nkv, kv_stride, dh;
  float scale;
  int nx;                         // workgroups per (batch row, head): launches are 1-D, nx * heads * batch
};

// Workgroup -> (tile, head, batch row).  Workgroup b runs on XCD b % 8 and every XCD has its own L2, so all nx
// workgroups of one (row, head) - which stream the same K / V (or Q / dO) panels - are placed on ONE XCD: pair
// = 8 * (slot / nx) + xcd.  With the plain x-fastest 3-D grid the 32 query tiles of a head were spread over all 8
// XCDs and each L2 fetched the panel again (rocprofv3 FETCH_SIZE: 428 MB per forward launch at 64x64, 2.5 x the
// algorithmic bytes).  Falls back to the plain order when heads * batch is not a multiple of 8.
struct BlkMap { int bx, h, b; };
__device__ __forceinline__ BlkMap attn_block_map(const AttnParams& p) {
  const int lid = blockIdx.x, nx = p.nx;
  const int pairs = p.heads * p.batch;
  int pair, bx;
  if ((pairs & 7) == 0) {
    const int xcd = lid & 7, slot = lid >> 3;
    const int grp = slot / nx;
    pair = grp * 8 + xcd;
    bx = slot - grp * nx;
  } else {
    pair = lid / nx;
    bx = lid - pair * nx;
  }
  const int b = pair / p.heads;
  return {bx, pair - b * p.heads, b};
}

constexpr float NEG_BIG = -1.0e30f;

#ifdef SKG_PHASES
// profiling build only (make phases; tools/attn_phases.py): per-wave cycle sums of the forward kernel's phases
__device__ unsigned long long g_attn_phase[1 << 16][8];
#define ATT_T(x) const unsigned long long x = __builtin_readcyclecounter()
#define ATT_ACC(i, a, b) att_acc[i] += (b) - (a)
#else
#define ATT_T(x) do { } while (0)
#define ATT_ACC(i, a, b) do { } while (0)
#endif

// max of three without the canonicalising v_max_f32 x, x, x that fmaxf() puts in front of every MFMA result (the
// compiler cannot prove an MFMA output is not a signalling NaN): 16 scores reduce in 8 VALU instead of 31
__device__ __forceinline__ float max3f(float a, float b, float c) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
  return r;
}
constexpr float LOG2E = 1.4426950408889634f;
constexpr float LN2 = 0.6931471805599453f;
constexpr int TP = 72;   // pitch (halves) of the transposed tiles [d][64 + 8]

// A-operand fragment of a transposed tile for k-step s: head-dim row (16 u + l16), keys by the map above
__device__ __forceinline__ half8_t tfrag(const half_t* tile, int u, int s, int l16, int g) {
  const half_t* p = tile + (u * 16 + l16) * TP + 32 * s + 4 * g;
  const half4_t lo = ld_half4(p), hi = ld_half4(p + 16);
  half8_t f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return f;
}
// The same fragment out of a ROW-MAJOR tile [key][VP] through the gfx950 transpose read: in a 16-lane group lane i hands
// in the address of 4 contiguous halves - row i >> 2, columns 4 (i & 3) .. + 3 of a [4][16] block - and receives column
// i of that block, rows 0 .. 3 (tools/ubench/tr_probe.hip).  Group g of fragment (u, s) takes the block of keys
// 32 s + 16 hh + 4 g .. + 3 x head-dim columns 16 u .. + 15, hh = 0 / 1: exactly the k <-> key map above, so V (and K, Q,
// dO in the backward kernels) needs no transposed copy in HBM.  `base` = tile + (4 g + (l16 >> 2)) * VP + 4 (l16 & 3).
// Banks: 8 key rows x 32 bytes per 32-lane group are conflict-free when VP = 16 (mod 32) halves.
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ half4_t tr_read(const half_t* p) {
  const fp16x4_t r = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(p));
  return __builtin_bit_cast(half4_t, r);
}
template <int VP>
__device__ __forceinline__ half8_t tfrag_rows(const half_t* base, int u, int s) {
  const half4_t lo = tr_read(base + (32 * s) * VP + 16 * u), hi = tr_read(base + (32 * s + 16) * VP + 16 * u);
  half8_t f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return f;
}
constexpr int vrow_pitch(int nd) { return nd * 16 + ((nd & 1) ? 0 : 16); }      // = 16 (mod 32) halves, >= 16 nd
// pack score-layout registers (4 tiles x 4) into the two B-operand fragments
__device__ __forceinline__ void pack_p(const float4_t (&s)[4], half8_t (&pb)[2]) {
#pragma unroll
  for (int k2 = 0; k2 < 2; ++k2)
#pragma unroll
    for (int i = 0; i < 8; ++i) pb[k2][i] = (half_t)s[2 * k2 + (i >> 2)][i & 3];
}

// register staging of [64][dh] row tiles (global -> VGPR now, VGPR -> LDS later)
template <int KS>
struct RowRegs { half8_t v[KS]; };               // 64 * KS*4 pieces / 256 threads

template <int KS>
__device__ __forceinline__ void rows_store(const RowRegs<KS>& r, half_t* __restrict__ dst) {
  constexpr int KP = KS * 32 + 16;      // pitch = 16 or 48 (mod 64) halves: conflict-free for the 4 x 16-lane groups of ds_read_b128 (+ 8 is 2-way)
  constexpr int PPR = KS * 4;
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    const int pi = threadIdx.x + q * 256;
    const int row = pi / PPR, pc = (pi - row * PPR) * 8;
    st_half8(dst + row * KP + pc, r.v[q]);
  }
}
// -------------------------------------------------------------------------------------------------
// register staging of one K tile ([64][dh] rows) and one V tile (transposed [dh][64], or row-major [64][VP] with VROW)
// per workgroup.  ONES (d = 40): the first padding row of V^T / column dh of the row-major tile is 1.0, so row dh of
// O^T = V^T P^T accumulates the softmax denominator sum_k p[k] on the matrix pipe instead of the VALU.
template <int KS, int ND>
struct KVRegs {
  static constexpr int NK = KS;                 // 64 * KS*4 pieces / 256 threads
  static constexpr int NV = (ND + 1) / 2;       // ND*16*8 pieces / 256 threads
  half8_t k[NK], v[NV];
};

template <int KS, int ND, bool VROW = false>
__device__ __forceinline__ void kv_store(const KVRegs<KS, ND>& r, half_t* __restrict__ Ks, half_t* __restrict__ Vs) {
  constexpr int KP = KS * 32 + 16;      // pitch = 16 or 48 (mod 64) halves: conflict-free for the 4 x 16-lane groups of ds_read_b128 (+ 8 is 2-way)
  constexpr int PPR = KS * 4;
#pragma unroll
  for (int q = 0; q < KVRegs<KS, ND>::NK; ++q) {
    const int pi = threadIdx.x + q * 256;
    const int row = pi / PPR, pc = (pi - row * PPR) * 8;
    st_half8(Ks + row * KP + pc, r.k[q]);
  }
#pragma unroll
  for (int q = 0; q < KVRegs<KS, ND>::NV; ++q) {
    const int pi = threadIdx.x + q * 256;
    if constexpr (VROW) {      // row-major V tile [64][VP]: 2 ND pieces per key row (a dense image when VP = 16 ND)
      constexpr int VP = vrow_pitch(ND);
      if (pi < ND * 128) st_half8(Vs + (pi / (2 * ND)) * VP + (pi % (2 * ND)) * 8, r.v[q]);
    } else {
      if (pi < ND * 128) st_half8(Vs + (pi >> 3) * TP + (pi & 7) * 8, r.v[q]);
    }
  }
}

// The same prefetch through buffer descriptors: every per-lane source offset is loop-invariant (row / column inside the
// tile, or an out-of-range constant for the zero-padded head-dim columns / rows), the tile position is a wave-uniform
// soffset and the descriptor's range check supplies the zeros behind the last key row - a prefetch is 2 * (KS + NV)
// instructions instead of ~15 VALU + 8 SALU per load of compare / select / zero-fill / exec masking (the round-2 counters:
// 140 VALU instructions per tile and wave of which only 32 + 16 + 20 are the softmax; VALU time ~ MFMA time at d = 40).
constexpr unsigned ATT_OOB = 0x80000000u;
template <int KS, int ND>
struct KVSrc {
  __amdgpu_buffer_rsrc_t rk, rv;
  unsigned ko[KVRegs<KS, ND>::NK], vo[KVRegs<KS, ND>::NV];
  int vcol[KVRegs<KS, ND>::NV];            // first key column of the lane's piece (ragged last tile only)
  unsigned ones;                           // bit q: piece q of this lane is the all-ones row (ONES)
};
template <int KS, int ND, bool ONES, bool VROW = false>
__device__ __forceinline__ KVSrc<KS, ND> kv_src(const half_t* Kb, int ldk, const half_t* Vb, int ldvt, int kvlim, int dh) {
  KVSrc<KS, ND> s;
  // K: rows of THIS batch row only (the next row's keys start right behind: the size is what cuts them off)
  s.rk = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (unsigned)(((size_t)(kvlim - 1) * ldk + dh) * 2), 0x00020000);
  s.rv = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, (unsigned)((VROW ? (size_t)(kvlim - 1) * ldvt + dh
                                                                           : (size_t)(dh - 1) * ldvt + kvlim) * 2), 0x00020000);
  constexpr int PPR = KS * 4;
  s.ones = 0;
#pragma unroll
  for (int q = 0; q < KVRegs<KS, ND>::NK; ++q) {
    const int pi = threadIdx.x + q * 256;
    const int row = pi / PPR, pc = (pi - row * PPR) * 8;
    s.ko[q] = pc < dh ? (unsigned)(row * ldk + pc) * 2u : ATT_OOB;
  }
#pragma unroll
  for (int q = 0; q < KVRegs<KS, ND>::NV; ++q) {
    const int pi = threadIdx.x + q * 256;
    if constexpr (VROW) {      // ldvt is V's row pitch here; rows behind the last key read as zero by the range check
      const int row = pi / (2 * ND), pc = (pi % (2 * ND)) * 8;
      s.vo[q] = (pi < ND * 128 && pc < dh) ? (unsigned)(row * ldvt + pc) * 2u : ATT_OOB;
      s.vcol[q] = 0;
      if (ONES && pi < ND * 128 && pc == dh) s.ones |= 1u << q;      // column dh of every key row = 1.0
    } else {
      const int d = pi >> 3, pc = (pi & 7) * 8;
      s.vo[q] = (pi < ND * 128 && d < dh) ? (unsigned)(d * ldvt + pc) * 2u : ATT_OOB;
      s.vcol[q] = pc;
      if (ONES && d == dh) s.ones |= 1u << q;
    }
  }
  return s;
}
__device__ __forceinline__ half8_t buf_half8(__amdgpu_buffer_rsrc_t r, unsigned voff, unsigned soff) {
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  const u4 v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
  return __builtin_bit_cast(half8_t, v);
}
template <int KS, int ND, bool ONES, bool VROW = false>
__device__ __forceinline__ void kv_load_buf(KVRegs<KS, ND>& r, const KVSrc<KS, ND>& s, int kv0, int ldk, int kvlim,
                                            int ldv = 0) {
  const unsigned sk = (unsigned)kv0 * (unsigned)ldk * 2u, sv = VROW ? (unsigned)kv0 * (unsigned)ldv * 2u : (unsigned)kv0 * 2u;
#pragma unroll
  for (int q = 0; q < KVRegs<KS, ND>::NK; ++q) r.k[q] = buf_half8(s.rk, s.ko[q], sk);
  if (VROW || kv0 + 64 <= kvlim) {         // wave-uniform
#pragma unroll
    for (int q = 0; q < KVRegs<KS, ND>::NV; ++q) r.v[q] = buf_half8(s.rv, s.vo[q], sv);
  } else {                                 // ragged last tile: key columns behind this batch row's keys read as zero
#pragma unroll
    for (int q = 0; q < KVRegs<KS, ND>::NV; ++q) r.v[q] = buf_half8(s.rv, kv0 + s.vcol[q] < kvlim ? s.vo[q] : ATT_OOB, sv);
  }
  if (ONES) {
    const half_t one = (half_t)1.f, zr = (half_t)0.f;
#pragma unroll
    for (int q = 0; q < KVRegs<KS, ND>::NV; ++q)
      if ((s.ones >> q) & 1u) r.v[q] = VROW ? half8_t{one, zr, zr, zr, zr, zr, zr, zr} : half8_t{one, one, one, one, one, one, one, one};
  }
}

// The backward kernels' row / transposed tiles through the same descriptors (see KVSrc)
template <int KS>
struct RowSrc { __amdgpu_buffer_rsrc_t r; unsigned o[KS]; };
template <int KS>
__device__ __forceinline__ RowSrc<KS> row_src(const half_t* src, int ld, int rlim, int dh) {
  RowSrc<KS> s;
  s.r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, (unsigned)(((size_t)(rlim - 1) * ld + dh) * 2), 0x00020000);
  constexpr int PPR = KS * 4;
#pragma unroll
  for (int q = 0; q < KS; ++q) {
    const int pi = threadIdx.x + q * 256;
    const int row = pi / PPR, pc = (pi - row * PPR) * 8;
    s.o[q] = pc < dh ? (unsigned)(row * ld + pc) * 2u : ATT_OOB;
  }
  return s;
}
template <int KS>
__device__ __forceinline__ void rows_load_buf(RowRegs<KS>& r, const RowSrc<KS>& s, int ld, int r0) {
  const unsigned so = (unsigned)r0 * (unsigned)ld * 2u;
#pragma unroll
  for (int q = 0; q < KS; ++q) r.v[q] = buf_half8(s.r, s.o[q], so);
}
// Forward.  Per 64-key tile: issue the global loads of the NEXT tile into registers, run S^T = K Q^T,
// online softmax and O^T += V^T P^T on the current LDS tile, then (barrier) spill the prefetched
// registers into LDS: the HBM/L2 latency of tile t+1 hides under the MFMA + VALU work of tile t.
// Softmax arithmetic is kept to ~6 VALU per score: raw v_exp_f32 (arguments are <= 0, no range fix-up),
// scale folded into one FMA, masking only on a ragged last tile.
// CAUSAL (the CLIP text encoder, modules/pipeline.py:55-57 -> transformers CLIPTextModel): key j is visible to
// query i only when j <= i; a separate instantiation so the UNet's kernels carry no extra test.
// VAR (experiments / per-shape tuning): bit 0 = request every fragment of a tile up front (PRE), bit 1 = one register
// prefetch set instead of two (frees 16 VGPRs), bits 2-3 = waves per SIMD the register allocation is bounded for (0 = default)
template <int KS, int ND, int QT, bool CAUSAL = false, int VAR = 0, bool VROW = false>
__global__ __launch_bounds__(256, ((VAR >> 2) ? (VAR >> 2) : KS >= 5 ? 1 : (KS == 2 && ND == 3 && !CAUSAL && QT == 2) ? 4 : 2)) void attn_fwd_kernel(const AttnParams p) {
  // QT query tiles of 16 per wave: a workgroup covers 64 * QT queries, so every K / V^T fragment read from LDS
  // (and every byte of K/V streamed from L2) is used by QT MFMAs instead of one.
  constexpr int KP = KS * 32 + 16;      // pitch = 16 or 48 (mod 64) halves: conflict-free for the 4 x 16-lane groups of ds_read_b128 (+ 8 is 2-way)
  constexpr int VP = vrow_pitch(ND);    // VROW: p.Vt / p.ldvt are the ROW-MAJOR V and its pitch; the tile is kept [64][VP] and read transposing
  constexpr int KSZ = 64 * KP, VSZ = VROW ? 64 * VP : ND * 16 * TP;
  __shared__ __attribute__((aligned(16))) half_t lds[2 * (KSZ + VSZ)];      // two (K, V^T) stages
  half_t* const Ks0 = lds;
  half_t* const Vs0 = lds + KSZ;
  half_t* const Ks1 = lds + KSZ + VSZ;
  half_t* const Vs1 = lds + 2 * KSZ + VSZ;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  const BlkMap bm = attn_block_map(p);
  const int b = bm.b, h = bm.h;
  const int dh = p.dh;
  int q[QT];
  bool qok[QT];
  half8_t qf[QT][KS];
#pragma unroll
  for (int i = 0; i < QT; ++i) {
    q[i] = bm.bx * (64 * QT) + (wave * QT + i) * 16 + l16;
    qok[i] = q[i] < p.Nq;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = 32 * ks + 8 * g;
      qf[i][ks] = (qok[i] && c < dh) ? ld_half8(p.Q + (size_t)(b * p.Nq + q[i]) * p.ldq + h * dh + c) : zero_half8();
    }
  }
  // Q is pre-multiplied by scale * log2(e) (one fp16 rounding, the size of Q's own) and MINUS the reference maximum
  // rides in as the C operand of the first QK^T MFMA, so a score leaves the matrix pipe as  s*sc - m  and the softmax
  // is exp2 + max + pack per score: no FMA, no cross-lane traffic.  The reference m only has to stay within 2^8 of the
  // true running maximum (p <= 256 is exact enough in fp16, O and the denominator share the reference), so it moves
  // when a tile beats it by more than 8 (rare after the first tile) instead of on every new maximum.
  const float sc = p.scale * LOG2E;
#pragma unroll
  for (int i = 0; i < QT; ++i)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[i][ks][j] = (half_t)((float)qf[i][ks][j] * sc);
  float4_t o[QT][ND];
  float4_t nm[QT];                  // minus the reference maximum (log2 domain), replicated: the MFMA C operand
  float l[QT];
#pragma unroll
  for (int i = 0; i < QT; ++i) {
    nm[i] = float4_t{0.f, 0.f, 0.f, 0.f}; l[i] = 0.f;
#pragma unroll
    for (int u = 0; u < ND; ++u) o[i][u] = float4_t{0.f, 0.f, 0.f, 0.f};
  }
  constexpr float REF_SLACK = 8.f;
  // (measured: requesting all fragments of a tile up front costs 14 VGPRs = the third wave per SIMD: 710 -> 767 us at
  // d = 40; the compiler's read-as-you-go order with three resident waves is faster.  Kept for experiments.)
  constexpr bool PRE = (VAR & 1) && KS <= 2 && 4 * KS * QT >= 2 * ND;
  constexpr int VREADS = 2;                                   // LDS instructions per V^T fragment (two ds_read_b64)
  // <2, 3> is dispatched for d = 40 only: 8 spare rows in the 48-row V^T tile -> the denominator comes out of the
  // PV MFMA (row 40 of O^T) and the 16 adds per tile and query tile leave the VALU, which bounds this head size
  constexpr bool ONES = (KS == 2 && ND == 3);

  const half_t* Kb = p.K + (size_t)b * p.kv_stride * p.ldk + h * dh;
  const half_t* Vb = VROW ? p.Vt + (size_t)b * p.kv_stride * p.ldvt + h * dh
                          : p.Vt + (size_t)h * dh * p.ldvt + (size_t)b * p.kv_stride;
  const int vlane = (4 * g + (l16 >> 2)) * VP + 4 * (l16 & 3);      // VROW: the lane's corner of every [4][16] block
  const int nt = (p.Nkv + 63) / 64;

#ifdef SKG_PHASES
  unsigned long long att_acc[6] = {0, 0, 0, 0, 0, 0};
#endif
  // one 64-key tile: S^T = K Q^T - m, online softmax, O^T += V^T P^T   (for the wave's QT query tiles)
  auto tile = [&](const half_t* __restrict__ Ks, const half_t* __restrict__ Vs, int kv0) {
    ATT_T(ta);
    float4_t s[QT][4];
    half8_t vf[ND][2];
    if constexpr (PRE) {
      // hipcc emits "2 ds_read -> s_waitcnt -> 4 MFMA" per 16 keys, i.e. one exposed LDS latency per group and the
      // V^T reads after the softmax.  Here every K fragment of the tile is requested up front and the V^T fragments
      // are requested under the QK^T MFMAs, so they land while the matrix pipe and then the softmax VALU work.
      half8_t kf[4][KS];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) kf[t][ks] = ld_half8(Ks + (16 * t + l16) * KP + 32 * ks + 8 * g);
#pragma unroll
      for (int u = 0; u < ND; ++u)
#pragma unroll
        for (int k2 = 0; k2 < 2; ++k2) vf[u][k2] = VROW ? tfrag_rows<VP>(Vs + vlane, u, k2) : tfrag(Vs, u, k2, l16, g);
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int i = 0; i < QT; ++i)
            s[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[t][ks], qf[i][ks], ks == 0 ? nm[i] : s[i][t], 0, 0, 0);
      __builtin_amdgcn_sched_group_barrier(0x100, 4 * KS, 0);            // all K fragment reads
#pragma unroll
      for (int r = 0; r < 2 * ND; ++r) {                                  // V^T reads woven under the first MFMAs
        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        __builtin_amdgcn_sched_group_barrier(0x100, VREADS, 0);
      }
      __builtin_amdgcn_sched_group_barrier(0x008, 4 * KS * QT - 2 * ND, 0);
    } else {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
          const half8_t kf = ld_half8(Ks + (16 * t + l16) * KP + 32 * ks + 8 * g);
#pragma unroll
          for (int i = 0; i < QT; ++i)
            s[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[i][ks], ks == 0 ? nm[i] : s[i][t], 0, 0, 0);
        }
      }
    }
    ATT_T(tb);
    ATT_ACC(0, ta, tb);                   // K fragment reads + QK^T MFMA issue
    const bool first = kv0 == 0;          // the reference starts at 0: the first tile always re-bases it
    half8_t pb[QT][2];
#pragma unroll
    for (int i = 0; i < QT; ++i) {
      if (CAUSAL || kv0 + 64 > p.Nkv) {          // ragged last tile only (wave-uniform)
        // key = kv0 + 4g + (16t + r): compare the compile-time part against per-lane limits computed in here, so
        // the full tiles carry no index arithmetic
        int lim = p.Nkv - kv0 - 4 * g;
        int qlim = CAUSAL ? q[i] - kv0 - 4 * g : 0;
        asm volatile("" : "+v"(lim), "+v"(qlim));      // keeps the 16 compares inside this (rare) branch
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * t + r >= lim || (CAUSAL && 16 * t + r > qlim)) s[i][t][r] = NEG_BIG;
      }
      const float m0 = max3f(s[i][0][0], s[i][0][1], s[i][0][2]);
      const float m1 = max3f(s[i][0][3], s[i][1][0], s[i][1][1]);
      const float m2 = max3f(s[i][1][2], s[i][1][3], s[i][2][0]);
      const float m3 = max3f(s[i][2][1], s[i][2][2], s[i][2][3]);
      const float m4 = max3f(s[i][3][0], s[i][3][1], s[i][3][2]);
      float mx = max3f(max3f(m0, m1, m2), max3f(m3, m4, s[i][3][3]), m0);
      if (__builtin_amdgcn_ballot_w64(first || mx > REF_SLACK) != 0) {      // wave-uniform, rare after tile 0
        mx = max3f(mx, __shfl_xor(mx, 16, 64), mx);
        mx = max3f(mx, __shfl_xor(mx, 32, 64), mx);                         // the query's maximum over the tile
        const float delta = (first || mx > REF_SLACK) ? mx : 0.f;
        const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);   // (o, l are still zero in tile 0)
        nm[i] -= delta;
        if (!ONES) l[i] *= alpha;
#pragma unroll
        for (int u = 0; u < ND; ++u) o[i][u] *= alpha;
#pragma unroll
        for (int t = 0; t < 4; ++t) s[i][t] -= delta;
      }
      float ps = 0.f;
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float e = __builtin_amdgcn_exp2f(s[i][t][r]);
          s[i][t][r] = e;
          if (!ONES) ps += e;
        }
      if (!ONES) l[i] += ps;
      pack_p(s[i], pb[i]);
    }
    ATT_T(tc);
    ATT_ACC(1, tb, tc);                   // wait for the scores + softmax VALU
#pragma unroll
    for (int u = 0; u < ND; ++u)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        if constexpr (!PRE) vf[u][k2] = VROW ? tfrag_rows<VP>(Vs + vlane, u, k2) : tfrag(Vs, u, k2, l16, g);
#pragma unroll
        for (int i = 0; i < QT; ++i) o[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[u][k2], pb[i][k2], o[i][u], 0, 0, 0);
      }
    ATT_T(td);
    ATT_ACC(2, tc, td);                   // V^T fragment reads + PV MFMA issue
  };

  // Two LDS stages (+ two register sets when they fit): tile t computes from stage t&1 while tile t+1 sits in
  // registers on its way to the other stage and tile t+2 is in flight from L2/HBM, ONE barrier per tile.
  // Invariant at the top of the (unrolled-by-2) loop, t even: stage 0 = tile t, r0 = tile t+1, r1 = tile t+2.
  constexpr bool DEEP = KS < 5 && !(KS == 2 && ND == 3) && !(VAR & 2);      // d = 160: a second register set would not fit 2 waves / SIMD; d = 40: four waves / SIMD instead
  const KVSrc<KS, ND> src = kv_src<KS, ND, ONES, VROW>(Kb, p.ldk, Vb, p.ldvt, p.kv_stride, dh);
  KVRegs<KS, ND> r0;
  kv_load_buf<KS, ND, ONES, VROW>(r0, src, 0, p.ldk, p.kv_stride, p.ldvt);
  kv_store<KS, ND, VROW>(r0, Ks0, Vs0);
  if (nt > 1) kv_load_buf<KS, ND, ONES, VROW>(r0, src, 64, p.ldk, p.kv_stride, p.ldvt);
  if constexpr (DEEP) {
    KVRegs<KS, ND> r1;
    if (nt > 2) kv_load_buf<KS, ND, ONES, VROW>(r1, src, 128, p.ldk, p.kv_stride, p.ldvt);
    __syncthreads();
    for (int t0 = 0; t0 < nt; t0 += 2) {
      tile(Ks0, Vs0, t0 * 64);
      if (t0 + 1 < nt) kv_store<KS, ND, VROW>(r0, Ks1, Vs1);
      if (t0 + 3 < nt) kv_load_buf<KS, ND, ONES, VROW>(r0, src, (t0 + 3) * 64, p.ldk, p.kv_stride, p.ldvt);
      __syncthreads();
      if (t0 + 1 < nt) {
        tile(Ks1, Vs1, (t0 + 1) * 64);
        if (t0 + 2 < nt) kv_store<KS, ND, VROW>(r1, Ks0, Vs0);
        if (t0 + 4 < nt) kv_load_buf<KS, ND, ONES, VROW>(r1, src, (t0 + 4) * 64, p.ldk, p.kv_stride, p.ldvt);
        __syncthreads();
      }
    }
  } else {
    // one register set: r0 = tile t+1 while tile t computes (prefetch distance one tile)
    __syncthreads();
    ATT_T(tl0);
    for (int t0 = 0; t0 < nt; t0 += 2) {
      tile(Ks0, Vs0, t0 * 64);
      ATT_T(t1);
      if (t0 + 1 < nt) kv_store<KS, ND, VROW>(r0, Ks1, Vs1);
      ATT_T(t2);
      __syncthreads();
      ATT_T(t3);
      ATT_ACC(3, t1, t2);                 // staging the next tile (waits for its global loads)
      ATT_ACC(4, t2, t3);                 // barrier
      if (t0 + 1 < nt) {
        if (t0 + 2 < nt) kv_load_buf<KS, ND, ONES, VROW>(r0, src, (t0 + 2) * 64, p.ldk, p.kv_stride, p.ldvt);
        tile(Ks1, Vs1, (t0 + 1) * 64);
        ATT_T(t4);
        if (t0 + 2 < nt) kv_store<KS, ND, VROW>(r0, Ks0, Vs0);
        if (t0 + 3 < nt) kv_load_buf<KS, ND, ONES, VROW>(r0, src, (t0 + 3) * 64, p.ldk, p.kv_stride, p.ldvt);
        ATT_T(t5);
        __syncthreads();
        ATT_T(t6);
        ATT_ACC(3, t4, t5);
        ATT_ACC(4, t5, t6);
      }
    }
    ATT_T(tl1);
    ATT_ACC(5, tl0, tl1);                 // whole loop
#ifdef SKG_PHASES
    if (lane == 0 && blockIdx.x < (1 << 14)) {
      for (int j = 0; j < 6; ++j) g_attn_phase[blockIdx.x * 4 + wave][j] = att_acc[j];
      g_attn_phase[blockIdx.x * 4 + wave][6] = nt;
    }
#endif
  }
#pragma unroll
  for (int i = 0; i < QT; ++i) {
    float li;
    if (ONES) {      // denominator = row dh = 40 of O^T: tile 2, row 8 -> lanes g == 2, element 0
      li = __shfl(o[i][ND - 1][0], 32 + l16, 64);
    } else {
      li = l[i];
      li += __shfl_xor(li, 16, 64);
      li += __shfl_xor(li, 32, 64);
    }
    const float inv = 1.f / li;
    if (qok[i]) {
      half_t* orow = p.O + (size_t)(b * p.Nq + q[i]) * p.ldo + h * dh;
#pragma unroll
      for (int u = 0; u < ND; ++u) {
        const int d = 16 * u + 4 * g;
        if (d < dh) {
          half4_t v = {(half_t)(o[i][u][0] * inv), (half_t)(o[i][u][1] * inv), (half_t)(o[i][u][2] * inv),
                       (half_t)(o[i][u][3] * inv)};
          st_half4(orow + d, v);
        }
      }
      if (p.lse && g == 0) p.lse[((size_t)b * p.heads + h) * p.Nq + q[i]] = (log2f(li) - nm[i][0]) * LN2;
    }
  }
}

// ---- forward for SHORT key sequences (cross-attention over the 77 text tokens: modules/pipeline.py:96 -> diffusers
// CrossAttention with encoder_hidden_states) -------------------------------------------------------------------------------
// Round 2 ran these through the flash kernel above: two 64-key tiles of which the second holds 13 keys, a prologue / epilogue
// per 128 queries that never amortises, 45 us at 64x64 for 14 us of HBM time (128-142 TFLOP/s).
// With <= 80 keys there is nothing to stream and nothing "online": K [80][d] and V [96][d] of one (row, head) are staged in
// LDS ONCE per workgroup (25 KB at d = 40: four workgroups per CU), every wave then walks 16-query tiles: Q fragments
// straight from global memory (the next tile's loads in flight while this one computes), S^T = K Q^T in 5 x KS MFMAs, one
// plain softmax over the wave's 20 scores per lane (the maximum is exact: no running reference), O^T = V^T P^T with the
// probabilities as the B operand out of the accumulator registers, normalise, store.  What is left is one read of Q and one
// write of O: the kernel is HBM-bound.
template <int KS, int ND>
__global__ __launch_bounds__(256, (KS >= 5 ? 2 : (KS >= 3 ? 3 : 4))) void attn_fwd_short_kernel(const AttnParams p, int qchunk) {
  constexpr int NT = 5;                 // 16-key score tiles: kv_stride <= 80
  constexpr int NSV = 3;                // 32-key k-steps of the PV product (the last half step is zero rows)
  constexpr int KP = KS * 32 + 16;
  constexpr int VP = vrow_pitch(ND);
  __shared__ __attribute__((aligned(16))) half_t Ks[16 * NT * KP];
  __shared__ __attribute__((aligned(16))) half_t Vs[32 * NSV * VP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  const BlkMap bm = attn_block_map(p);
  const int b = bm.b, h = bm.h, dh = p.dh;
  const half_t* Kb = p.K + (size_t)b * p.kv_stride * p.ldk + h * dh;
  const half_t* Vb = p.Vt + (size_t)b * p.kv_stride * p.ldvt + h * dh;
  // ---- stage K [80][KP] and V [96][VP]: rows behind the last key and head-dim columns behind dh are zero
  for (int pi = threadIdx.x; pi < 16 * NT * (KS * 4); pi += 256) {
    const int row = pi / (KS * 4), pc = (pi - row * (KS * 4)) * 8;
    st_half8(Ks + row * KP + pc, (row < p.Nkv && pc < dh) ? ld_half8(Kb + (size_t)row * p.ldk + pc) : zero_half8());
  }
  for (int pi = threadIdx.x; pi < 32 * NSV * (2 * ND); pi += 256) {
    const int row = pi / (2 * ND), pc = (pi - row * (2 * ND)) * 8;
    st_half8(Vs + row * VP + pc, (row < p.Nkv && pc < dh) ? ld_half8(Vb + (size_t)row * p.ldvt + pc) : zero_half8());
  }
  __syncthreads();
  const int vlane = (4 * g + (l16 >> 2)) * VP + 4 * (l16 & 3);
  const float sc = p.scale * LOG2E;
  // keys this lane holds in score tile t, register r: 16 t + 4 g + r.  77 text tokens: only the last tile is ragged
  // (three registers of the g = 3 lanes); fewer than 65 keys (tests, other callers) take the general mask
  bool dead[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) dead[r] = 16 * (NT - 1) + 4 * g + r >= p.Nkv;
  const bool short_rows = p.Nkv <= 16 * (NT - 1);        // wave-uniform
  const int klim = p.Nkv - 4 * g;
  const int q_begin = bm.bx * qchunk, q_end = min(p.Nq, q_begin + qchunk);
  const size_t rowbase = (size_t)b * p.Nq;
  auto load_q = [&](half8_t (&dst)[KS], int q0) {          // rows behind q_end read row q_end - 1 (not stored)
    const int q = min(q0 + l16, q_end - 1);
    const half_t* src = p.Q + (rowbase + q) * p.ldq + h * dh + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) dst[ks] = (32 * ks + 8 * g < dh) ? ld_half8(src + 32 * ks) : zero_half8();
  };
  half8_t qn[KS];
  int q0 = q_begin + wave * 16;
  if (q0 < q_end) load_q(qn, q0);
  for (; q0 < q_end; q0 += 64) {
    half8_t qf[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[ks][j] = (half_t)((float)qn[ks][j] * sc);
    if (q0 + 64 < q_end) load_q(qn, q0 + 64);               // in flight under this tile's work
    // (the accumulators start from an opaque zero REGISTER, the operands of an accumulator-starting MFMA stay live past
    // it, and the k-steps of one accumulator are separated by the other tiles' MFMAs - the treatment xattn.hip got after its
    // first build lost half an accumulator; tools/ubench/mfma_fresh_overlap.hip could not make the pattern fail in isolation,
    // profiles/r04_mfma_fresh_overlap.txt, so this is belt and braces)
    float4_t s[NT];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const half8_t kf = ld_half8(Ks + (16 * t + l16) * KP + 32 * ks + 8 * g);
        if (ks == 0) {
          float4_t z = {0.f, 0.f, 0.f, 0.f};
          asm volatile("" : "+v"(z));
          s[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[ks], z, 0, 0, 0);
          asm volatile("" ::"v"(kf), "v"(qf[ks]));
        } else {
          s[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf, qf[ks], s[t], 0, 0, 0);
        }
      }
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (dead[r]) s[NT - 1][r] = NEG_BIG;
    if (short_rows) {
#pragma unroll
      for (int t = 0; t < NT - 1; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (16 * t + r >= klim) s[t][r] = NEG_BIG;
    }
    float mx = max3f(s[0][0], s[0][1], s[0][2]);
    mx = max3f(mx, s[0][3], s[1][0]); mx = max3f(mx, s[1][1], s[1][2]); mx = max3f(mx, s[1][3], s[2][0]);
    mx = max3f(mx, s[2][1], s[2][2]); mx = max3f(mx, s[2][3], s[3][0]); mx = max3f(mx, s[3][1], s[3][2]);
    mx = max3f(mx, s[3][3], s[4][0]); mx = max3f(mx, s[4][1], s[4][2]); mx = max3f(mx, s[4][3], mx);
    mx = max3f(mx, __shfl_xor(mx, 16, 64), mx);
    mx = max3f(mx, __shfl_xor(mx, 32, 64), mx);
    float li = 0.f;
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float e = __builtin_amdgcn_exp2f(s[t][r] - mx);
        s[t][r] = e;
        li += e;
      }
    li += __shfl_xor(li, 16, 64);
    li += __shfl_xor(li, 32, 64);
    // B-operand fragments of P^T: element i of k-step sv <-> key 32 sv + 16 (i >> 2) + 4 g + (i & 3)
    half8_t pb[NSV];
#pragma unroll
    for (int sv = 0; sv < NSV; ++sv)
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const int t = 2 * sv + (i >> 2);
        pb[sv][i] = t < NT ? (half_t)s[t][i & 3] : (half_t)0.f;
      }
    float4_t o[ND];
#pragma unroll
    for (int sv = 0; sv < NSV; ++sv)
#pragma unroll
      for (int u = 0; u < ND; ++u) {
        const half8_t vfr = tfrag_rows<VP>(Vs + vlane, u, sv);
        if (sv == 0) {
          float4_t z = {0.f, 0.f, 0.f, 0.f};
          asm volatile("" : "+v"(z));
          o[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vfr, pb[sv], z, 0, 0, 0);
          asm volatile("" ::"v"(vfr), "v"(pb[sv]));
        } else {
          o[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vfr, pb[sv], o[u], 0, 0, 0);
        }
      }
    const int q = q0 + l16;
    if (q < q_end) {
      const float inv = 1.f / li;
      half_t* orow = p.O + (rowbase + q) * p.ldo + h * dh;
#pragma unroll
      for (int u = 0; u < ND; ++u) {
        const int d = 16 * u + 4 * g;
        if (d < dh) {
          half4_t v = {(half_t)(o[u][0] * inv), (half_t)(o[u][1] * inv), (half_t)(o[u][2] * inv), (half_t)(o[u][3] * inv)};
          st_half4(orow + d, v);
        }
      }
      if (p.lse && g == 0) p.lse[((size_t)b * p.heads + h) * p.Nq + q] = (log2f(li) + mx) * LN2;
    }
  }
}

#if defined(SKG_LAB) || defined(SKG_PHASES)      // withdrawn round-4 formulation (+ 2.5 % on the kernel, - 3 % on config 5): lab / probe builds only
// ---- forward, 8 waves, 32 queries per wave, MFMA and softmax of DIFFERENT tiles in one instruction stream (round 4) --------
// attn_fwd_kernel above runs QK^T -> softmax -> PV of one tile as one dependency chain per wave and hopes that another wave's
// MFMAs land under this wave's softmax; the counters say they do not (matrix pipe 0.43 busy at d = 40: 448 MFMA + ~316 VALU
// cycles per tile and wave against ~1 040 measured, EXPERIMENTS.md round 3).  Measured on the way to this kernel (EXPERIMENTS.md
// round 4): a lone wave issues one VALU instruction per ~8.5 cycles whatever the instruction, so a softmax SEGMENT of ~100
// instructions is ~1 500 cycles long and a barrier-separated ping-pong (MFMA segment | softmax segment, as gemm8.hip) is bound by
// it (784 us against 608).  What a wave can do is put its MFMAs INTO that cadence: here every iteration issues
//     O^T += V(t-1)^T P(t-1)^T      and      S(t+1)^T = K(t+1) Q^T - m      (18 MFMAs at d = 40, 384 matrix-pipe cycles)
// beside  P(t) = exp2(S(t))  (VALU) - three different tiles, no dependency between them inside the iteration (S and P are double
// buffered in registers) - so the matrix pipe works under the wave's own softmax and the partner wave of the SIMD fills the rest.
// Formulation (swapped products as above, other instruction shapes):
//   * S^T[64 keys x 32 queries] = K Q^T with v_mfma_f32_32x32x16_f16: K = 16 steps, so d = 40 pads to 48 (3 steps) instead of
//     64 and every K fragment read feeds a 32 x 32 tile.  In the C layout lane (q = lane & 31, hi = lane >> 5) holds
//     S^T[key = 32 kt + (r & 3) + 8 (r >> 2) + 4 hi][q] in register r of key tile kt: 32 scores of ONE query per lane.
//   * O^T[d x 16 queries] += V^T P^T stays on v_mfma_f32_16x16x32_f16 (d pads to 16 ND, not to a multiple of 32).  Its B
//     operand wants lane (n = lane & 15, g = lane >> 4) = query n of a 16-query tile; the packed probabilities sit in lanes
//     (q = lane & 31): four v_permlane16_swap_b32 per key tile (rows 1, 3 of the first operand <-> rows 0, 2 of the second)
//     turn {regs 0..7, regs 8..15} of all 32 queries into the B operands of the two 16-query tiles.  k-slot (g, i) of a 32-key
//     step then means key (i & 3) + 8 (i >> 2) + 4 (g >> 1) + 16 (g & 1); the V tile is staged with key bits 2 and 4 swapped so
//     that the transposing LDS read of group g starts at row 16 (g >> 1) + 4 (g & 1) (+ 8 for its second half) - the same
//     conflict-free pattern as tfrag_rows.
//   * online softmax as above: scale folded into Q, minus the reference maximum as the MFMA C operand, re-based only when a
//     tile beats it by more than 2^8 (wave-uniform ballot): at a re-base everything still at the old reference - O, the
//     denominator AND the probabilities of tile t that wait for their PV - is scaled exactly once (cdna_hip_programming.md T13).
//     d = 40: the denominator is row 40 of O^T (ones column of V).
// K / V tiles: registers -> LDS, one barrier per tile: tile t + 2 is written in iteration t (K ring of 2: K(t + 2) replaces
// K(t), last read in iteration t - 1; V ring of 4: V(t + 2) replaces V(t - 2), and V(t - 1) is being read), its loads were issued in
// iteration t - 1; pitches kp8_pitch (odd number of 16-byte pieces: conflict-free ds_read_b128 with row = lane & 31) / vrow_pitch.
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));

constexpr int kp8_pitch(int d16) { return (2 * d16) | 1; }      // 16-byte pieces per K row: >= 2 d16, odd

// NW waves per workgroup (32 NW queries), OCC = waves per SIMD the register allocation is bounded for: <8, 2> = one 8-wave
// workgroup per CU; <4, 3> = three independent 4-wave workgroups per CU (their phases drift apart by themselves)
// PROBE (profiling build only, make phases + SKG_ATTN8_PROBE; results are WRONG): 1 = no exponentials, 2 = no MFMAs, 4 = no staging,
// 8 = no barrier, 16 = no LDS fragment reads
template <int D16, int ND, bool ONES, int NW = 8, int OCC = 2, int PROBE = 0>
__global__ __launch_bounds__(64 * NW, OCC) void attn_fwd8_kernel(const AttnParams p) {
  constexpr int NTHR = 64 * NW;
  // DEEP (the 8-wave form: 256 registers per wave): every latency is covered by software pipelining instead of by other waves -
  // the fragments of iteration t + 1 are read from LDS during iteration t (two register sets), tile t + 3 is written in
  // iteration t and its global loads were issued two iterations earlier (two staging sets)
  constexpr bool DEEP = NW == 8;
  constexpr int NK = 2, NV = 4;                      // ring depths
  constexpr int KP = kp8_pitch(D16) * 8;             // halves
  constexpr int VP = vrow_pitch(ND);
  constexpr int KSZ = 64 * KP, VSZ = 64 * VP, VBASE = NK * KSZ;
  __shared__ __attribute__((aligned(16))) half_t lds[NK * KSZ + NV * VSZ];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l32 = lane & 31, hi = lane >> 5, l16 = lane & 15, g = lane >> 4;
  const BlkMap bm = attn_block_map(p);
  const int b = bm.b, h = bm.h, dh = p.dh;
  const int PK = dh >> 3;                            // 16-byte pieces per K / V row in memory
  const int q0 = bm.bx * (32 * NW) + wave * 32;

  // ---- Q^T fragments (B operand of 32x32x16: lane (n = l32, hi) holds d = 16 ks + 8 hi .. + 7 of query n), pre-scaled
  const int qrow = min(q0 + l32, p.Nq - 1);
  half8_t qf[D16];
  {
    const half_t* qp = p.Q + (size_t)(b * p.Nq + qrow) * p.ldq + h * dh + 8 * hi;
    const float sc = p.scale * LOG2E;
#pragma unroll
    for (int ks = 0; ks < D16; ++ks) {
      qf[ks] = (16 * ks + 8 * hi < dh) ? ld_half8(qp + 16 * ks) : zero_half8();
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[ks][j] = (half_t)((float)qf[ks][j] * sc);
    }
  }

  // ---- staging: piece pi of a tile = (K | V, key row, 16-byte column piece); a thread owns pieces tid + NTHR j
  constexpr int MAXPC = (2 * 64 * 2 * ND + NTHR - 1) / NTHR;       // >= 2 * 64 * PK / 512 for every head width this instantiation takes
  const half_t* Kb = p.K + (size_t)b * p.kv_stride * p.ldk + h * dh;
  const half_t* Vb = p.Vt + (size_t)b * p.kv_stride * p.ldvt + h * dh;
  const __amdgpu_buffer_rsrc_t rk = __builtin_amdgcn_make_buffer_rsrc((void*)Kb, 0, (unsigned)(((size_t)(p.kv_stride - 1) * p.ldk + dh) * 2), 0x00020000);
  const __amdgpu_buffer_rsrc_t rv = __builtin_amdgcn_make_buffer_rsrc((void*)Vb, 0, (unsigned)(((size_t)(p.kv_stride - 1) * p.ldvt + dh) * 2), 0x00020000);
  unsigned soff[MAXPC];      // byte offset inside a tile in memory, or ATT_OOB
  int sdst[MAXPC];           // half offset inside its ring slot, or -1
  bool sisv[MAXPC];          // wave-uniform: 64 * PK is a multiple of 64
#pragma unroll
  for (int j = 0; j < MAXPC; ++j) {
    const int pi = tid + NTHR * j;
    const int npk = 64 * PK;
    const bool isv = pi >= npk;
    const int pj = isv ? pi - npk : pi;
    const int row = pj / PK, pc = pj - row * PK;
    const bool ok = pi < 2 * npk;
    sisv[j] = __builtin_amdgcn_readfirstlane((int)isv) != 0;
    const int vrow = (row & ~20) | ((row & 4) << 2) | ((row & 16) >> 2);      // V rows: key bits 2 and 4 swapped (see above)
    soff[j] = ok ? (unsigned)(row * (isv ? p.ldvt : p.ldk) + pc * 8) * 2u : ATT_OOB;
    sdst[j] = ok ? (isv ? vrow * VP + pc * 8 : row * KP + pc * 8) : -1;
  }
  half8_t sreg[MAXPC], sreg2[DEEP ? MAXPC : 1];
  auto stage_load_to = [&](half8_t (&r)[MAXPC], int t) {      // global -> registers (rows behind the last key of this batch row read as zero)
    const unsigned sk = (unsigned)t * 64u * (unsigned)p.ldk * 2u, sv = (unsigned)t * 64u * (unsigned)p.ldvt * 2u;
#pragma unroll
    for (int j = 0; j < MAXPC; ++j) r[j] = sisv[j] ? buf_half8(rv, soff[j], sv) : buf_half8(rk, soff[j], sk);
  };
  auto stage_store_from = [&](const half8_t (&r)[MAXPC], int t) {     // registers -> K ring slot t % NK / V ring slot t % NV
    half_t* const ks = lds + (t & (NK - 1)) * KSZ;
    half_t* const vs = lds + VBASE + (t & (NV - 1)) * VSZ;
#pragma unroll
    for (int j = 0; j < MAXPC; ++j)
      if (sdst[j] >= 0) st_half8((sisv[j] ? vs : ks) + sdst[j], r[j]);
  };
  auto stage_load = [&](int t) { stage_load_to(sreg, t); };
  auto stage_store = [&](int t) { stage_store_from(sreg, t); };
  // constant pieces of every ring slot, written once: the zero columns d >= dh of K that the last k-step reads, and (ONES) the
  // piece [1, 0 x 7] at column dh of every V row - the register staging never touches them
  for (int i = tid; i < (NK + NV) * 64; i += NTHR) {
    const int slot = i >> 6, row = i & 63;
    if (slot < NK) {
      for (int c = PK * 8; c < D16 * 16; c += 8) st_half8(lds + slot * KSZ + row * KP + c, zero_half8());
    } else {
      half_t* const vr = lds + VBASE + (slot - NK) * VSZ + row * VP;
      if (ONES) {
        const half8_t one = {(half_t)1.f, 0, 0, 0, 0, 0, 0, 0};
        st_half8(vr + PK * 8, one);
      } else {
        for (int c = PK * 8; c < ND * 16; c += 8) st_half8(vr + c, zero_half8());
      }
    }
  }
  const int nt = (p.Nkv + 63) >> 6;
  stage_load(0);
  stage_store(0);
  if (nt > 1) { stage_load(1); stage_store(1); }
  if (nt > 2) stage_load(2);             // (DEEP: written after K(0)'s fragments have been read, below)

  // ---- per-wave state
  float16_t sA[2], sB[2];            // S^T of two tiles in flight: key tiles 0, 1
  uint4_t pA[2][2], pB[2][2];        // P^T of two tiles as B operands: [query tile][32-key step]
  float4_t o[2][ND];                 // O^T: query tile (queries 0-15, 16-31 of the wave) x d tile
  float nm = 0.f, l = 0.f;           // minus the reference maximum (log2 domain) / denominator partial (S layout: query l32)
#pragma unroll
  for (int qt = 0; qt < 2; ++qt)
#pragma unroll
    for (int u = 0; u < ND; ++u) o[qt][u] = float4_t{0.f, 0.f, 0.f, 0.f};
  constexpr float REF_SLACK = 8.f;
  const int vlane = (16 * (g >> 1) + 4 * (g & 1) + (l16 >> 2)) * VP + 4 * (l16 & 3);
  const int klane = l32 * KP + 8 * hi;

  // fragment reads of one iteration, all issued up front (two waves per SIMD: nobody else covers an exposed LDS latency)
  auto read_k = [&](int t, half8_t (&kf)[2][D16]) {
    const half_t* Ks = lds + (t & (NK - 1)) * KSZ + klane;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int ks = 0; ks < D16; ++ks) kf[kt][ks] = (PROBE & 16) ? qf[ks] : ld_half8(Ks + 32 * kt * KP + 16 * ks);
  };
  auto read_v = [&](int t, half8_t (&vf)[ND][2]) {
    const half_t* Vs = lds + VBASE + (t & (NV - 1)) * VSZ + vlane;
#pragma unroll
    for (int u = 0; u < ND; ++u)
#pragma unroll
      for (int sv = 0; sv < 2; ++sv) {
        if (PROBE & 16) { vf[u][sv] = qf[(u + sv) % D16]; continue; }
        const half4_t lo = tr_read(Vs + (32 * sv) * VP + 16 * u), hh = tr_read(Vs + (32 * sv + 8) * VP + 16 * u);
        vf[u][sv] = half8_t{lo[0], lo[1], lo[2], lo[3], hh[0], hh[1], hh[2], hh[3]};
      }
  };
  // S^T = K Q^T - m: the two key tiles alternate (dependent MFMAs one apart)
  auto qk = [&](const half8_t (&kf)[2][D16], float16_t (&s)[2]) {
    if (PROBE & 2) { asm volatile("" : "+v"(s[0]), "+v"(s[1])); return; }
    const float16_t init = {nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm, nm};
#pragma unroll
    for (int ks = 0; ks < D16; ++ks)
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
        s[kt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf[kt][ks], qf[ks], ks == 0 ? init : s[kt], 0, 0, 0);
  };
  // O^T += V^T P^T
  auto pv = [&](const half8_t (&vf)[ND][2], const uint4_t (&pb)[2][2]) {
    if (PROBE & 2) return;
#pragma unroll
    for (int sv = 0; sv < 2; ++sv)
#pragma unroll
      for (int u = 0; u < ND; ++u)
#pragma unroll
        for (int qt = 0; qt < 2; ++qt)
          o[qt][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf[u][sv], __builtin_bit_cast(half8_t, pb[qt][sv]), o[qt][u], 0, 0, 0);
  };
  // P = exp2(S) -> packed fp16 -> B operands of the two 16-query tiles
  auto expo = [&](const float16_t (&s)[2], uint4_t (&pb)[2][2]) {
    float ps = 0.f;
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) {
      unsigned pk[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float e0 = (PROBE & 1) ? s[kt][2 * j] : __builtin_amdgcn_exp2f(s[kt][2 * j]);
        const float e1 = (PROBE & 1) ? s[kt][2 * j + 1] : __builtin_amdgcn_exp2f(s[kt][2 * j + 1]);
        if (!ONES) ps += e0 + e1;
        const half2_t h2 = {(half_t)e0, (half_t)e1};
        pk[j] = __builtin_bit_cast(unsigned, h2);
      }
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const auto r = __builtin_amdgcn_permlane16_swap(pk[j], pk[j + 4], false, false);
        pb[0][kt][j] = r[0];
        pb[1][kt][j] = r[1];
      }
    }
    if (!ONES) l += ps;
  };
  // the lane's maximum over its 32 scores of tile t (ragged last tile masked first)
  auto tile_max = [&](auto is_last, int t, float16_t (&s)[2]) {
    const int kv0 = t * 64;
    if (decltype(is_last)::value && kv0 + 64 > p.Nkv) {      // ragged last tile (only the iterations without a next tile test for it)
      int lim = p.Nkv - kv0 - 4 * hi;
      asm volatile("" : "+v"(lim));
#pragma unroll
      for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          if (32 * kt + (r & 3) + 8 * (r >> 2) >= lim) s[kt][r] = NEG_BIG;
    }
    float m8[8];
#pragma unroll
    for (int j = 0; j < 5; ++j) m8[j] = max3f(s[0][3 * j], s[0][3 * j + 1], s[0][3 * j + 2]);
    m8[5] = max3f(s[0][15], s[1][0], s[1][1]);
    m8[6] = max3f(s[1][2], s[1][3], s[1][4]);
    m8[7] = max3f(s[1][5], s[1][6], s[1][7]);
    float mx = max3f(max3f(m8[0], m8[1], m8[2]), max3f(m8[3], m8[4], m8[5]), max3f(m8[6], m8[7], s[1][8]));
    mx = max3f(mx, max3f(s[1][9], s[1][10], s[1][11]), max3f(s[1][12], s[1][13], s[1][14]));
    return max3f(mx, s[1][15], mx);
  };
  // the reference moves (rare after tile 0; wave-uniform): O and the denominator hold every tile before t - in program order
  // the PV MFMAs of tile t - 1 are BEHIND us here - so nothing else is at the old reference (cdna_hip_programming.md T13)
  auto rebase = [&](bool first, float mx, float16_t (&s)[2]) {
    mx = max3f(mx, __shfl_xor(mx, 32, 64), mx);                           // the query's maximum over the tile
    const float delta = (first || mx > REF_SLACK) ? mx : 0.f;
    const float alpha = first ? 1.f : __builtin_amdgcn_exp2f(-delta);     // (o, l are still zero in tile 0)
    nm -= delta;
    l *= alpha;
    const float aA = __shfl(alpha, l16, 64), aB = __shfl(alpha, 16 + l16, 64);        // O layout: query l16 of each 16-query tile
#pragma unroll
    for (int u = 0; u < ND; ++u) { o[0][u] *= aA; o[1][u] *= aB; }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt) s[kt] -= delta;
  };
  // One iteration = tile t's softmax on the VALU beside the MFMAs of two OTHER tiles:
  //   block 1:  O^T += V(t-1)^T P(t-1)^T   ||  lane maxima of S(t)          (then the rare re-base)
  //   block 2:  S(t+1)^T = K(t+1) Q^T - m  ||  P(t) = exp2(S(t)), packing, relayout
#ifdef SKG_PHASES
  unsigned long long att_acc[6] = {0, 0, 0, 0, 0, 0};
#endif
  auto step = [&](auto has_pv, auto has_qk, int t, float16_t (&sc)[2], float16_t (&sn)[2], uint4_t (&pp)[2][2], uint4_t (&pc)[2][2]) {
    constexpr bool HAS_PV = decltype(has_pv)::value, HAS_QK = decltype(has_qk)::value;
    ATT_T(ta);
    half8_t vf[ND][2], kf[2][D16];
    if constexpr (HAS_PV) read_v(t - 1, vf);
    if constexpr (HAS_QK) read_k(t + 1, kf);
    if constexpr (HAS_PV) pv(vf, pp);
    const float mx = tile_max(std::integral_constant<bool, !HAS_QK>{}, t, sc);
    ATT_T(tb);
    if (__builtin_amdgcn_ballot_w64(t == 0 || mx > REF_SLACK) != 0) rebase(t == 0, mx, sc);
    ATT_T(tc);
    if constexpr (HAS_QK) qk(kf, sn);
    expo(sc, pc);
    ATT_T(td);
    if (!(PROBE & 4)) {
      if (t + 2 < nt) stage_store(t + 2);
      if (t + 3 < nt) stage_load(t + 3);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    ATT_T(te);
    __builtin_amdgcn_sched_barrier(0);
    if (!(PROBE & 8)) asm volatile("s_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    ATT_T(tf);
    ATT_ACC(0, ta, tb); ATT_ACC(1, tb, tc); ATT_ACC(2, tc, td); ATT_ACC(3, td, te); ATT_ACC(4, te, tf); ATT_ACC(5, ta, tf);
  };
  using T_ = std::true_type;
  using F_ = std::false_type;
  if constexpr (DEEP) {
    half8_t kfA[2][D16], kfB[2][D16], vfA[ND][2], vfB[ND][2];      // fragments of K(j) / V(j) live in set j & 1
    // iteration t: MFMAs PV(t - 1) [vfu, pp] and QK(t + 1) [kfu]  ||  softmax of S(t)  ||  fragment reads V(t) -> vff, K(t + 2) -> kff
    // ||  tile t + 3 -> LDS from staging set sst, loads of tile t + 5 into it
    auto dstep = [&](auto has_pv, auto has_qk, int t, float16_t (&sc)[2], float16_t (&sn)[2], uint4_t (&pp)[2][2], uint4_t (&pc)[2][2],
                     half8_t (&kfu)[2][D16], half8_t (&kff)[2][D16], half8_t (&vfu)[ND][2], half8_t (&vff)[ND][2], half8_t (&sst)[MAXPC]) {
      constexpr bool HAS_PV = decltype(has_pv)::value, HAS_QK = decltype(has_qk)::value;
      ATT_T(ta);
      if (t + 2 < nt) read_k(t + 2, kff);
      read_v(t, vff);
      if constexpr (HAS_PV) pv(vfu, pp);
      const float mx = tile_max(std::integral_constant<bool, !HAS_QK>{}, t, sc);
      ATT_T(tb);
      if (__builtin_amdgcn_ballot_w64(t == 0 || mx > REF_SLACK) != 0) rebase(t == 0, mx, sc);
      ATT_T(tc);
      if constexpr (HAS_QK) qk(kfu, sn);
      expo(sc, pc);
      ATT_T(td);
      if (!(PROBE & 4)) {
        if (t + 3 < nt) stage_store_from(sst, t + 3);
        if (t + 5 < nt) stage_load_to(sst, t + 5);
      }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      ATT_T(te);
      __builtin_amdgcn_sched_barrier(0);
      if (!(PROBE & 8)) asm volatile("s_barrier" ::: "memory");
      __builtin_amdgcn_sched_barrier(0);
      ATT_T(tf);
      ATT_ACC(0, ta, tb); ATT_ACC(1, tb, tc); ATT_ACC(2, tc, td); ATT_ACC(3, td, te); ATT_ACC(4, te, tf); ATT_ACC(5, ta, tf);
    };
#define DSTEP_EVEN(PV_, QK_, t_) dstep(PV_{}, QK_{}, t_, sA, sB, pB, pA, kfB, kfA, vfB, vfA, sreg2)
#define DSTEP_ODD(PV_, QK_, t_) dstep(PV_{}, QK_{}, t_, sB, sA, pA, pB, kfA, kfB, vfA, vfB, sreg)
    __syncthreads();                    // tiles 0, 1 and the constant pieces are in LDS
    read_k(0, kfA);
    qk(kfA, sA);
    if (nt > 1) read_k(1, kfB);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");     // K(0)'s slot takes K(2)
    if (nt > 2) stage_store(2);
    if (nt > 3) stage_load_to(sreg2, 3);      // tile j travels in staging set (j & 1): sreg2 = odd tiles
    if (nt > 4) stage_load_to(sreg, 4);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (nt > 1) DSTEP_EVEN(F_, T_, 0); else DSTEP_EVEN(F_, F_, 0);
    int t = 1;
    for (; t + 2 < nt; t += 2) {
      DSTEP_ODD(T_, T_, t);
      DSTEP_EVEN(T_, T_, t + 1);
    }
    if (nt > 1) {
      if (t + 1 < nt) {
        DSTEP_ODD(T_, T_, t);
        DSTEP_EVEN(T_, F_, t + 1);
        pv(vfA, pA);                     // tile nt - 1 (even): its V fragments were read in the last iteration
      } else {
        DSTEP_ODD(T_, F_, t);
        pv(vfB, pB);
      }
    } else {
      pv(vfA, pA);
    }
#undef DSTEP_EVEN
#undef DSTEP_ODD
  } else {

  __syncthreads();                      // tiles 0, 1 and the constant pieces are in LDS
  {
    half8_t kf[2][D16];
    read_k(0, kf);
    qk(kf, sA);
  }
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // K(0)'s slot is refilled in iteration 0
  if (nt > 1) step(F_{}, T_{}, 0, sA, sB, pB, pA); else step(F_{}, F_{}, 0, sA, sB, pB, pA);
  int t = 1;                             // t odd: S / P of tile t are sB / pB
  for (; t + 2 < nt; t += 2) {
    step(T_{}, T_{}, t, sB, sA, pA, pB);
    step(T_{}, T_{}, t + 1, sA, sB, pB, pA);
  }
  bool last_in_a = true;                 // where P of the last tile ends up
  if (nt > 1) {
    if (t + 1 < nt) {                    // t = nt - 2 (odd): one more full iteration, then the last tile (even)
      step(T_{}, T_{}, t, sB, sA, pA, pB);
      step(T_{}, F_{}, t + 1, sA, sB, pB, pA);
    } else {                             // t = nt - 1 (odd)
      step(T_{}, F_{}, t, sB, sA, pA, pB);
      last_in_a = false;
    }
  }
  {
    half8_t vf[ND][2];
    read_v(nt - 1, vf);
    if (last_in_a) pv(vf, pA); else pv(vf, pB);
  }

  }
#ifdef SKG_PHASES
  if (lane == 0 && blockIdx.x < (1 << 13)) {
    for (int j = 0; j < 6; ++j) g_attn_phase[blockIdx.x * NW + wave][j] = att_acc[j];
    g_attn_phase[blockIdx.x * NW + wave][6] = nt;
  }
#endif
  // ---- epilogue: O^T tiles -> O rows (lane (l16, g): d = 16 u + 4 g .. + 3 of query l16 of each 16-query tile)
  float liA, liB;
  if (ONES) {      // denominator = row dh = 40 of O^T: tile 2, row 8 -> lanes g == 2, element 0
    liA = __shfl(o[0][ND - 1][0], 32 + l16, 64);
    liB = __shfl(o[1][ND - 1][0], 32 + l16, 64);
  } else {
    const float ls = l + __shfl_xor(l, 32, 64);
    liA = __shfl(ls, l16, 64);
    liB = __shfl(ls, 16 + l16, 64);
  }
#pragma unroll
  for (int qt = 0; qt < 2; ++qt) {
    const int q = q0 + 16 * qt + l16;
    const float inv = 1.f / (qt ? liB : liA);
    if (q < p.Nq) {
      half_t* orow = p.O + (size_t)(b * p.Nq + q) * p.ldo + h * dh;
#pragma unroll
      for (int u = 0; u < ND; ++u) {
        const int d = 16 * u + 4 * g;
        if (d < dh) {
          const half4_t v = {(half_t)(o[qt][u][0] * inv), (half_t)(o[qt][u][1] * inv), (half_t)(o[qt][u][2] * inv), (half_t)(o[qt][u][3] * inv)};
          st_half4(orow + d, v);
        }
      }
    }
  }
  if (p.lse) {       // S layout: lanes 0..31 own query q0 + lane
    const float li = (lane & 16) ? liB : liA;
    if (hi == 0 && q0 + l32 < p.Nq) p.lse[((size_t)b * p.heads + h) * p.Nq + q0 + l32] = (log2f(li) - nm) * LN2;
  }
}

#endif      // SKG_LAB || SKG_PHASES

// dQ: per 64-query block, loop over key tiles.  dS^T = P^T o (dP^T - delta);  dQ^T += K^T dS^T.
template <int KS, int ND, int QT>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_kernel(const AttnParams p) {
  // QT query tiles of 16 per wave (64 * QT queries per workgroup): each K / V / K^T fragment read feeds QT MFMAs
  constexpr int KP = KS * 32 + 16;      // pitch = 16 or 48 (mod 64) halves: conflict-free for the 4 x 16-lane groups of ds_read_b128 (+ 8 is 2-way)
  __shared__ __attribute__((aligned(16))) half_t Ks[64 * KP];
  __shared__ __attribute__((aligned(16))) half_t Vr[64 * KP];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  const BlkMap bm = attn_block_map(p);
  const int b = bm.b, h = bm.h;
  const int dh = p.dh;
  int q[QT];
  bool qok[QT];
  half8_t qf[QT][KS], dof[QT][KS];
  float4_t nlse[QT], ndl[QT];
  float4_t dq[QT][ND];
#pragma unroll
  for (int i = 0; i < QT; ++i) {
    q[i] = bm.bx * (64 * QT) + (wave * QT + i) * 16 + l16;
    qok[i] = q[i] < p.Nq;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = 32 * ks + 8 * g;
      const bool ok = qok[i] && c < dh;
      qf[i][ks] = ok ? ld_half8(p.Q + (size_t)(b * p.Nq + q[i]) * p.ldq + h * dh + c) : zero_half8();
      dof[i][ks] = ok ? ld_half8(p.dO + (size_t)(b * p.Nq + q[i]) * p.lddo + h * dh + c) : zero_half8();
    }
    const size_t sidx = ((size_t)b * p.heads + h) * p.Nq + (qok[i] ? q[i] : 0);
    // minus lse (log2 domain) and minus delta ride in as the C operands of the S and dP MFMAs (see the forward)
    float dl;
    if (p.Of) {      // (launch-uniform) delta = sum_c O dO of this query from the dO fragments already in registers: the four lanes of a
                     // query hold 8 channels of every 32 each; the sum also goes to delta_out for the dK / dV launch
      float sm = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int c = 32 * ks + 8 * g;
        if (qok[i] && c < dh) {
          const half8_t of = ld_half8(p.Of + (size_t)(b * p.Nq + q[i]) * p.ldof + h * dh + c);
#pragma unroll
          for (int j = 0; j < 8; ++j) sm += (float)of[j] * (float)dof[i][ks][j];
        }
      }
      sm += __shfl_xor(sm, 16, 64);
      sm += __shfl_xor(sm, 32, 64);
      if (g == 0 && qok[i]) p.delta_out[sidx] = sm;
      dl = sm;
    } else {
      dl = qok[i] ? p.delta[sidx] : 0.f;
    }
    const float nl = qok[i] ? -p.lse[sidx] * LOG2E : NEG_BIG, nd = qok[i] ? -dl : 0.f;
    nlse[i] = float4_t{nl, nl, nl, nl};
    ndl[i] = float4_t{nd, nd, nd, nd};
#pragma unroll
    for (int u = 0; u < ND; ++u) dq[i][u] = float4_t{0.f, 0.f, 0.f, 0.f};
  }
  const float sc = p.scale * LOG2E;
#pragma unroll
  for (int i = 0; i < QT; ++i)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) qf[i][ks][j] = (half_t)((float)qf[i][ks][j] * sc);   // same rounding as the forward

  const half_t* Kb = p.K + (size_t)b * p.kv_stride * p.ldk + h * dh;
  const half_t* Vb = p.V + (size_t)b * p.kv_stride * p.ldv + h * dh;
  const int nt = (p.Nkv + 63) / 64;
  // register prefetch: the two tiles of key block t+1 are loaded while block t computes.  K^T fragments (the A operand
  // of dQ^T += K^T dS^T) come out of the SAME row-major K tile through the LDS transpose read (tfrag_rows).
  RowRegs<KS> rk, rv;
  const RowSrc<KS> sk = row_src<KS>(Kb, p.ldk, p.kv_stride, dh), sv = row_src<KS>(Vb, p.ldv, p.kv_stride, dh);
  const int tlane = (4 * g + (l16 >> 2)) * KP + 4 * (l16 & 3);
  rows_load_buf<KS>(rk, sk, p.ldk, 0);
  rows_load_buf<KS>(rv, sv, p.ldv, 0);
  for (int t0 = 0; t0 < nt; ++t0) {
    const int kv0 = t0 * 64;
    __syncthreads();                     // everyone is done reading the previous block
    rows_store<KS>(rk, Ks);
    rows_store<KS>(rv, Vr);
    __syncthreads();
    if (t0 + 1 < nt) {
      rows_load_buf<KS>(rk, sk, p.ldk, kv0 + 64);
      rows_load_buf<KS>(rv, sv, p.ldv, kv0 + 64);
    }
    float4_t s[QT][4], dp[QT][4];      // s = S*sc - lse,  dp = dP - delta  straight out of the matrix pipe
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int off = (16 * t + l16) * KP + 32 * ks + 8 * g;
        const half8_t kfr = ld_half8(Ks + off), vfr = ld_half8(Vr + off);
#pragma unroll
        for (int i = 0; i < QT; ++i) {
          s[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kfr, qf[i][ks], ks == 0 ? nlse[i] : s[i][t], 0, 0, 0);
          dp[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vfr, dof[i][ks], ks == 0 ? ndl[i] : dp[i][t], 0, 0, 0);
        }
      }
    }
    half8_t sb[QT][2];
#pragma unroll
    for (int i = 0; i < QT; ++i) {
      if (kv0 + 64 > p.Nkv) {          // ragged last block only (wave-uniform)
        const int lim = p.Nkv - kv0 - 4 * g;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            if (16 * t + r >= lim) s[i][t][r] = NEG_BIG;
      }
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) s[i][t][r] = __builtin_amdgcn_exp2f(s[i][t][r]) * dp[i][t][r];
      pack_p(s[i], sb[i]);
    }
#pragma unroll
    for (int u = 0; u < ND; ++u)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const half8_t ktf = tfrag_rows<KP>(Ks + tlane, u, k2);
#pragma unroll
        for (int i = 0; i < QT; ++i) dq[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ktf, sb[i][k2], dq[i][u], 0, 0, 0);
      }
  }
#pragma unroll
  for (int i = 0; i < QT; ++i) {
    if (!qok[i]) continue;
    half_t* orow = p.O + (size_t)(b * p.Nq + q[i]) * p.ldo + h * dh;
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int d = 16 * u + 4 * g;
      if (d < dh) {
        half4_t v = {(half_t)(dq[i][u][0] * p.scale), (half_t)(dq[i][u][1] * p.scale), (half_t)(dq[i][u][2] * p.scale),
                     (half_t)(dq[i][u][3] * p.scale)};
        st_half4(orow + d, v);
      }
    }
  }
}

// dK/dV: per 64-key block (lane column = key), loop over query tiles.
//   S = Q K^T (rows q), P = exp(S*scale - lse[q]), dP = dO V^T, dS = P o (dP - delta[q])
//   dV^T += dO^T P,   dK^T += Q^T dS      (A operands = transposed tiles, B = registers)
template <int KS, int ND, int KT>
__global__ __launch_bounds__(256, (KS >= 5 ? 1 : 2)) void attn_bwd_dkv_kernel(const AttnParams p) {
  // KT key tiles of 16 per wave: a workgroup owns 64 * KT keys, so every Q / dO / Q^T / dO^T fragment read from LDS
  // (and every byte of those panels streamed from L2) feeds KT MFMAs instead of one (same idea as QT in the forward).
  constexpr int KP = KS * 32 + 16;      // pitch = 16 or 48 (mod 64) halves: conflict-free for the 4 x 16-lane groups of ds_read_b128 (+ 8 is 2-way)
  __shared__ __attribute__((aligned(16))) half_t Qs[64 * KP];
  __shared__ __attribute__((aligned(16))) half_t Ds[64 * KP];
  __shared__ __attribute__((aligned(16))) float lse_s[64], del_s[64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l16 = lane & 15, g = lane >> 4;
  const BlkMap bm = attn_block_map(p);
  const int b = bm.b, h = bm.h;
  const int dh = p.dh;
  int kv[KT];
  bool kok[KT];
  half8_t kf[KT][KS], vf[KT][KS];
#pragma unroll
  for (int i = 0; i < KT; ++i) {
    kv[i] = bm.bx * (64 * KT) + (wave * KT + i) * 16 + l16;
    kok[i] = kv[i] < p.Nkv;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      const int c = 32 * ks + 8 * g;
      const bool ok = kok[i] && c < dh;
      kf[i][ks] = ok ? ld_half8(p.K + (size_t)(b * p.kv_stride + kv[i]) * p.ldk + h * dh + c) : zero_half8();
      vf[i][ks] = ok ? ld_half8(p.V + (size_t)(b * p.kv_stride + kv[i]) * p.ldv + h * dh + c) : zero_half8();
    }
  }
  const float sc = p.scale * LOG2E;
#pragma unroll
  for (int i = 0; i < KT; ++i)
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
      for (int j = 0; j < 8; ++j) kf[i][ks][j] = (half_t)((float)kf[i][ks][j] * sc);   // S*sc out of the matrix pipe
  float4_t dk[KT][ND], dv[KT][ND];
#pragma unroll
  for (int i = 0; i < KT; ++i)
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      dk[i][u] = float4_t{0.f, 0.f, 0.f, 0.f};
      dv[i][u] = float4_t{0.f, 0.f, 0.f, 0.f};
    }
  const half_t* Qb = p.Q + (size_t)b * p.Nq * p.ldq + h * dh;
  const half_t* Db = p.dO + (size_t)b * p.Nq * p.lddo + h * dh;
  const size_t sbase = ((size_t)b * p.heads + h) * p.Nq;
  const int nt = (p.Nq + 63) / 64;
  // register prefetch of query block t+1 (Q, dO row tiles, lse, delta) under block t's math; the Q^T / dO^T fragments
  // of dK^T += Q^T dS, dV^T += dO^T P are read from the same row tiles through the LDS transpose read
  RowRegs<KS> rq, rd;
  float r_lse = 0.f, r_del = 0.f;
  const RowSrc<KS> sq = row_src<KS>(Qb, p.ldq, p.Nq, dh), sd = row_src<KS>(Db, p.lddo, p.Nq, dh);
  const int tlane = (4 * g + (l16 >> 2)) * KP + 4 * (l16 & 3);
  auto prefetch = [&](int q0) {
    rows_load_buf<KS>(rq, sq, p.ldq, q0);
    rows_load_buf<KS>(rd, sd, p.lddo, q0);
    if (threadIdx.x < 64) {
      const int qq = q0 + threadIdx.x;
      r_lse = qq < p.Nq ? -p.lse[sbase + qq] * LOG2E : NEG_BIG;      // stored NEGATED: they are MFMA C operands
      r_del = qq < p.Nq ? -p.delta[sbase + qq] : 0.f;
    }
  };
  prefetch(0);
  for (int t0 = 0; t0 < nt; ++t0) {
    const int q0 = t0 * 64;
    __syncthreads();
    rows_store<KS>(rq, Qs);
    rows_store<KS>(rd, Ds);
    if (threadIdx.x < 64) { lse_s[threadIdx.x] = r_lse; del_s[threadIdx.x] = r_del; }
    __syncthreads();
    if (t0 + 1 < nt) prefetch(q0 + 64);
    float4_t s[KT][4], dp[KT][4];      // s = S*sc - lse[q],  dp = dP - delta[q]  (rows 16t + 4g + r are queries)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const float4_t nl = *reinterpret_cast<const float4_t*>(lse_s + 16 * t + 4 * g);
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const int off = (16 * t + l16) * KP + 32 * ks + 8 * g;
        const half8_t qfr = ld_half8(Qs + off), dfr = ld_half8(Ds + off);
#pragma unroll
        for (int i = 0; i < KT; ++i) {
          s[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qfr, kf[i][ks], ks == 0 ? nl : s[i][t], 0, 0, 0);
          dp[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(dfr, vf[i][ks], ks == 0 ? float4_t{0.f, 0.f, 0.f, 0.f} : dp[i][t], 0, 0, 0);
        }
      }
    }
    half8_t pb[KT][2], sb[KT][2];
#pragma unroll
    for (int i = 0; i < KT; ++i) {
      float4_t ds[4];
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pr = __builtin_amdgcn_exp2f(s[i][t][r]);
          s[i][t][r] = pr;
          ds[t][r] = pr * (dp[i][t][r] + del_s[16 * t + 4 * g + r]);
        }
      pack_p(s[i], pb[i]);
      pack_p(ds, sb[i]);
    }
#pragma unroll
    for (int u = 0; u < ND; ++u)
#pragma unroll
      for (int k2 = 0; k2 < 2; ++k2) {
        const half8_t dtf = tfrag_rows<KP>(Ds + tlane, u, k2), qtf = tfrag_rows<KP>(Qs + tlane, u, k2);
#pragma unroll
        for (int i = 0; i < KT; ++i) {
          dv[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(dtf, pb[i][k2], dv[i][u], 0, 0, 0);
          dk[i][u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(qtf, sb[i][k2], dk[i][u], 0, 0, 0);
        }
      }
  }
#pragma unroll
  for (int i = 0; i < KT; ++i) {
    if (!kok[i]) continue;
    half_t* krow = p.O + (size_t)(b * p.kv_stride + kv[i]) * p.ldo + h * dh;
    half_t* vrow = p.O2 + (size_t)(b * p.kv_stride + kv[i]) * p.ldo2 + h * dh;
#pragma unroll
    for (int u = 0; u < ND; ++u) {
      const int d = 16 * u + 4 * g;
      if (d < dh) {
        half4_t a = {(half_t)(dk[i][u][0] * p.scale), (half_t)(dk[i][u][1] * p.scale),
                     (half_t)(dk[i][u][2] * p.scale), (half_t)(dk[i][u][3] * p.scale)};
        half4_t c = {(half_t)dv[i][u][0], (half_t)dv[i][u][1], (half_t)dv[i][u][2], (half_t)dv[i][u][3]};
        st_half4(krow + d, a);
        st_half4(vrow + d, c);
      }
    }
  }
}

// delta[b][h][q] = sum_d dO[q][h*dh+d] * O[q][h*dh+d]; one 16-lane group per (row, head)
__global__ __launch_bounds__(256) void attn_delta_kernel(const half_t* __restrict__ O, int ldo,
                                                         const half_t* __restrict__ dO, int lddo,
                                                         float* __restrict__ delta, int batch, int heads, int Nq,
                                                         int dh) {
  const size_t total = (size_t)batch * Nq * heads;
  const size_t item = ((size_t)blockIdx.x * 256 + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  float s = 0.f;
  if (item < total) {
    const int h = (int)(item % heads);
    const size_t row = item / heads;   // b*Nq + q
    for (int c = sub * 8; c < dh; c += 128) {
      const half8_t a = ld_half8(O + row * ldo + h * dh + c);
      const half8_t d = ld_half8(dO + row * lddo + h * dh + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) s += (float)a[j] * (float)d[j];
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (item < total && sub == 0) {
    const int h = (int)(item % heads);
    const size_t row = item / heads;
    const size_t b = row / Nq, q = row - b * Nq;
    delta[(b * heads + h) * Nq + q] = s;
  }
}

// d = 40 / d = 64 forward: how many 16-query tiles a wave carries (QT) against how many waves a SIMD holds.
// Every K / V fragment a wave reads from LDS feeds QT MFMAs, and at QT = 2 the 16 resident waves of a CU move 56 KB through the
// LDS per 448 MFMA cycles - half the LDS pipe's time, which the probes of round 4 priced at 135 of 632 us at d = 40.  More tiles
// per wave trade that traffic for occupancy (the register budget): measured on one box (profiles/r04_attn_qt_variants.txt)
//   d = 40:  QT 2, 4 waves / SIMD 529 us;  QT 3, 3 waves / SIMD (165 VGPRs) 514-520 us;  QT 4, 2 waves / SIMD 554-562 us
//   d = 64:  QT 2, 3 waves / SIMD 1053 us at 9 480 tokens;  QT 4, 2 waves / SIMD (256 VGPRs) 1001-1006 us there, but
//            218 against 209 us at 2 568 tokens and 71 against 57 us at 840 (too few workgroups of 256 queries)
// End to end (same box, alternating): config 2 + 0.3 % / + 0.1 % with QT 3 at d = 40; config 5 - 0.4 % / + 0.0 % with QT 4 at d = 64.
// So QT 3 runs d = 40 where the launch has >= 1024 workgroups of 192 queries, d = 64 stays at QT 2 with ONE register prefetch
// set (152 VGPRs = three waves per SIMD instead of two: + 8...15 %).  SKG_ATTN_VAR: 12 = QT 2 everywhere (round 3's
// dispatch), 9 / 8 = QT 3 at d = 40 / QT 4 at d = 64 whatever the size, 7 = the two-set form at d = 64 (tools/attn_var_bench.py).
inline int attn_var();
template <int KS, int ND, bool VROW>
static void attn_fwd_launch_qt(AttnParams& p, hipStream_t st) {
  const int var = attn_var();
  const long rows = (long)p.heads * p.batch;
  if constexpr (ND == 3) {
    if (var == 9 || (var == 0 && skg_cdiv(p.Nq, 192) * rows >= 1024)) {
      p.nx = skg_cdiv(p.Nq, 192);
      hipLaunchKernelGGL((attn_fwd_kernel<KS, ND, 3, false, 2 | (3 << 2), VROW>), dim3((unsigned)(p.nx * rows)), dim3(256), 0, st, p);
      return;
    }
  } else if constexpr (VROW) {      // (the transposed-V form of this instantiation spills 5 registers: row-major V only)
    if (var == 8) {      // (opt-in: + 5 % on the kernel alone at 9 480 tokens, nothing end to end on config 5)
      p.nx = skg_cdiv(p.Nq, 256);
      hipLaunchKernelGGL((attn_fwd_kernel<KS, ND, 4, false, 2 | (2 << 2), VROW>), dim3((unsigned)(p.nx * rows)), dim3(256), 0, st, p);
      return;
    }
  }
  p.nx = skg_cdiv(p.Nq, 128);
  const dim3 grid((unsigned)(p.nx * rows));
  if (ND == 4 && var != 7)
    hipLaunchKernelGGL((attn_fwd_kernel<KS, ND, 2, false, 2 | (3 << 2), VROW>), grid, dim3(256), 0, st, p);
  else
    hipLaunchKernelGGL((attn_fwd_kernel<KS, ND, 2, false, 0, VROW>), grid, dim3(256), 0, st, p);
}
#define SKG_ATTN_VARIANTS(KS_, ND_, grid2, VROW_) attn_fwd_launch_qt<KS_, ND_, VROW_>(p, st)

// forward: two query tiles per wave (128 queries per workgroup) except at d = 160 (register budget)
#define SKG_ATTN_FWD_DISPATCH(grid1, grid2, VROW_)                                                                 \
  switch (p.dh) {                                                                                                  \
    case 16: hipLaunchKernelGGL((attn_fwd_kernel<1, 1, 2, false, 0, VROW_>), grid2, dim3(256), 0, st, p); break;   \
    case 32: hipLaunchKernelGGL((attn_fwd_kernel<1, 2, 2, false, 0, VROW_>), grid2, dim3(256), 0, st, p); break;   \
    case 40: SKG_ATTN_VARIANTS(2, 3, grid2, VROW_); break;                                                         \
    case 64: SKG_ATTN_VARIANTS(2, 4, grid2, VROW_); break;                                                         \
    case 80: hipLaunchKernelGGL((attn_fwd_kernel<3, 5, 2, false, 0, VROW_>), grid2, dim3(256), 0, st, p); break;   \
    case 160: hipLaunchKernelGGL((attn_fwd_kernel<5, 10, 1, false, 0, VROW_>), grid1, dim3(256), 0, st, p); break; \
    default: return SKG_E_UNSUPPORTED;                                                                             \
  }

#define SKG_ATTN_FWD_CAUSAL_DISPATCH(grid2)                                                               \
  switch (p.dh) {                                                                                        \
    case 16: hipLaunchKernelGGL((attn_fwd_kernel<1, 1, 2, true>), grid2, dim3(256), 0, st, p); break;    \
    case 32: hipLaunchKernelGGL((attn_fwd_kernel<1, 2, 2, true>), grid2, dim3(256), 0, st, p); break;    \
    case 64: hipLaunchKernelGGL((attn_fwd_kernel<2, 4, 2, true>), grid2, dim3(256), 0, st, p); break;    \
    default: return SKG_E_UNSUPPORTED;                                                                   \
  }

inline int attn_var() {
  static const int v = getenv("SKG_ATTN_VAR") ? atoi(getenv("SKG_ATTN_VAR")) : 0;
  return v;
}

inline bool common_ok(int batch, int heads, int Nq, int Nkv, int kv_stride, int dh) {
  return batch > 0 && heads > 0 && Nq > 0 && Nkv > 0 && kv_stride >= Nkv && kv_stride % 8 == 0 && dh % 8 == 0;
}

}  // namespace

static int attn_fwd_impl(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O,
                         int ldo, float* lse, int batch, int heads, int Nq, int Nkv, int kv_stride, int dh,
                         float scale, bool causal, void* stream, bool vrow = false) {
  SKG_REQUIRE(Q && K && Vt && O && common_ok(batch, heads, Nq, Nkv, kv_stride, dh));
  SKG_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldvt % 8 == 0 && ldo % 4 == 0);
  SKG_REQUIRE(skg_aligned(Q, 16) && skg_aligned(K, 16) && skg_aligned(Vt, 16) && skg_aligned(O, 8));
  AttnParams p{};
  p.Q = (const half_t*)Q; p.ldq = ldq; p.K = (const half_t*)K; p.ldk = ldk; p.Vt = (const half_t*)Vt; p.ldvt = ldvt;
  p.O = (half_t*)O; p.ldo = ldo; p.lse = lse;
  p.batch = batch; p.heads = heads; p.Nq = Nq; p.Nkv = Nkv; p.kv_stride = kv_stride; p.dh = dh; p.scale = scale;
  hipStream_t st = (hipStream_t)stream;
  // short key sequences (the 77 text tokens of every cross-attention): the LDS-resident kernel, HBM-bound
  static const bool no_short = getenv("SKG_NO_ATTN_SHORT") != nullptr;        // A/B switch
  if (vrow && !causal && kv_stride <= 80 && !no_short && (dh == 40 || dh == 64 || dh == 80 || dh == 160)) {
    // queries per workgroup: ~1024+ workgroups where the launch has them, at least 64, a multiple of 64
    long qch = ((long)heads * batch * Nq / 1024 + 63) / 64 * 64;
    qch = qch < 64 ? 64 : (qch > 512 ? 512 : qch);
    p.nx = skg_cdiv(Nq, (int)qch);
    dim3 gs((unsigned)p.nx * heads * batch);
    switch (dh) {
      case 40: hipLaunchKernelGGL((attn_fwd_short_kernel<2, 3>), gs, dim3(256), 0, st, p, (int)qch); break;
      case 64: hipLaunchKernelGGL((attn_fwd_short_kernel<2, 4>), gs, dim3(256), 0, st, p, (int)qch); break;
      case 80: hipLaunchKernelGGL((attn_fwd_short_kernel<3, 5>), gs, dim3(256), 0, st, p, (int)qch); break;
      default: hipLaunchKernelGGL((attn_fwd_short_kernel<5, 10>), gs, dim3(256), 0, st, p, (int)qch); break;
    }
    SKG_CHECK_LAUNCH("skg_attn_fwd (short keys)");
    return SKG_OK;
  }
#if defined(SKG_LAB) || defined(SKG_PHASES)
  // lab / probe builds only: the withdrawn 8-wave formulation (attn_fwd8_kernel) for the self-attention of the 64 x 64 / 32 x 32
  // levels, on request (SKG_ATTN8 = 1: one 8-wave workgroup per CU, 2: three 4-wave ones; SKG_NO_ATTN8 overrides)
  static const bool no8 = getenv("SKG_NO_ATTN8") != nullptr || getenv("SKG_ATTN8") == nullptr;
  static const int form8 = getenv("SKG_ATTN8") ? atoi(getenv("SKG_ATTN8")) : 0;      // 1: one 8-wave workgroup per CU, 2: three 4-wave ones
  if (vrow && !causal && !no8 && (dh == 40 || dh == 64) && (long)skg_cdiv(Nq, 256) * heads * batch >= 192) {
    const int qpw = form8 == 2 ? 128 : 256;
    p.nx = skg_cdiv(Nq, qpw);
    dim3 g8((unsigned)p.nx * heads * batch);
#ifdef SKG_PHASES
    static const int probe8 = getenv("SKG_ATTN8_PROBE") ? atoi(getenv("SKG_ATTN8_PROBE")) : 0;
    if (probe8 && dh == 40 && form8 == 2) {
      switch (probe8) {
        case 1: hipLaunchKernelGGL((attn_fwd8_kernel<3, 3, true, 4, 3, 1>), g8, dim3(256), 0, st, p); break;
        case 2: hipLaunchKernelGGL((attn_fwd8_kernel<3, 3, true, 4, 3, 2>), g8, dim3(256), 0, st, p); break;
        case 4: hipLaunchKernelGGL((attn_fwd8_kernel<3, 3, true, 4, 3, 4>), g8, dim3(256), 0, st, p); break;
        case 8: hipLaunchKernelGGL((attn_fwd8_kernel<3, 3, true, 4, 3, 8>), g8, dim3(256), 0, st, p); break;
        case 12: hipLaunchKernelGGL((attn_fwd8_kernel<3, 3, true, 4, 3, 12>), g8, dim3(256), 0, st, p); break;
        case 16: hipLaunchKernelGGL((attn_fwd8_kernel<3, 3, true, 4, 3, 16>), g8, dim3(256), 0, st, p); break;
        case 28: hipLaunchKernelGGL((attn_fwd8_kernel<3, 3, true, 4, 3, 28>), g8, dim3(256), 0, st, p); break;
        case 29: hipLaunchKernelGGL((attn_fwd8_kernel<3, 3, true, 4, 3, 29>), g8, dim3(256), 0, st, p); break;
        default: hipLaunchKernelGGL((attn_fwd8_kernel<3, 3, true, 4, 3, 30>), g8, dim3(256), 0, st, p); break;
      }
      SKG_CHECK_LAUNCH("skg_attn_fwd (8 waves, probe)");
      return SKG_OK;
    }
#endif
    if (form8 == 2) {
      if (dh == 40) hipLaunchKernelGGL((attn_fwd8_kernel<3, 3, true, 4, 3>), g8, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((attn_fwd8_kernel<4, 4, false, 4, 2>), g8, dim3(256), 0, st, p);
    } else {
      if (dh == 40) hipLaunchKernelGGL((attn_fwd8_kernel<3, 3, true>), g8, dim3(512), 0, st, p);
      else hipLaunchKernelGGL((attn_fwd8_kernel<4, 4, false>), g8, dim3(512), 0, st, p);
    }
    SKG_CHECK_LAUNCH("skg_attn_fwd (8 waves)");
    return SKG_OK;
  }
#endif
  p.nx = skg_cdiv(Nq, dh == 160 ? 64 : 128);       // query tiles per workgroup: see SKG_ATTN_FWD_DISPATCH
  dim3 grid((unsigned)p.nx * heads * batch);
  if (causal) {
    SKG_REQUIRE(dh != 160 && !vrow);
    SKG_ATTN_FWD_CAUSAL_DISPATCH(grid);
  } else if (vrow) {
    SKG_ATTN_FWD_DISPATCH(grid, grid, true);
  } else {
    SKG_ATTN_FWD_DISPATCH(grid, grid, false);
  }
  SKG_CHECK_LAUNCH("skg_attn_fwd");
  return SKG_OK;
}

extern "C" int skg_attn_fwd(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O,
                            int ldo, float* lse, int batch, int heads, int Nq, int Nkv, int kv_stride, int dh,
                            float scale, void* stream) {
  return attn_fwd_impl(Q, ldq, K, ldk, Vt, ldvt, O, ldo, lse, batch, heads, Nq, Nkv, kv_stride, dh, scale, false, stream);
}

// V handed over ROW-MAJOR ([batch * kv_stride][ldv], e.g. the third column block of a fused QKV projection): the kernel
// reads its fragments through the LDS transpose read, no V^T copy is needed
extern "C" int skg_attn_fwd_rowv(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, void* O,
                                 int ldo, float* lse, int batch, int heads, int Nq, int Nkv, int kv_stride, int dh,
                                 float scale, void* stream) {
  return attn_fwd_impl(Q, ldq, K, ldk, V, ldv, O, ldo, lse, batch, heads, Nq, Nkv, kv_stride, dh, scale, false, stream, true);
}

extern "C" int skg_attn_fwd_causal(const void* Q, int ldq, const void* K, int ldk, const void* Vt, int ldvt, void* O,
                                   int ldo, float* lse, int batch, int heads, int Nq, int Nkv, int kv_stride, int dh,
                                   float scale, void* stream) {
  return attn_fwd_impl(Q, ldq, K, ldk, Vt, ldvt, O, ldo, lse, batch, heads, Nq, Nkv, kv_stride, dh, scale, true, stream);
}

extern "C" int skg_attn_bwd_delta(const void* O, int ldo, const void* dO, int lddo, float* delta, int batch,
                                  int heads, int Nq, int dh, void* stream) {
  SKG_REQUIRE(O && dO && delta && batch > 0 && heads > 0 && Nq > 0 && dh % 8 == 0 && ldo % 8 == 0 && lddo % 8 == 0);
  SKG_REQUIRE(skg_aligned(O, 16) && skg_aligned(dO, 16));
  const size_t total = (size_t)batch * Nq * heads;
  hipLaunchKernelGGL(attn_delta_kernel, dim3((unsigned)((total * 16 + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, (const half_t*)O, ldo, (const half_t*)dO, lddo, delta, batch, heads, Nq, dh);
  SKG_CHECK_LAUNCH("skg_attn_bwd_delta");
  return SKG_OK;
}

static int attn_bwd_dq_impl(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* dO, int lddo,
                            const float* lse, const float* delta, void* dQ, int lddq, int batch, int heads, int Nq, int Nkv,
                            int kv_stride, int dh, float scale, void* stream, const void* O, int ldo, float* delta_out) {
  SKG_REQUIRE(Q && K && V && dO && lse && (delta || (O && delta_out)) && dQ && common_ok(batch, heads, Nq, Nkv, kv_stride, dh));
  SKG_REQUIRE(!O || (ldo % 8 == 0 && skg_aligned(O, 16)));
  SKG_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && lddo % 8 == 0 && lddq % 4 == 0);
  SKG_REQUIRE(skg_aligned(Q, 16) && skg_aligned(K, 16) && skg_aligned(V, 16) && skg_aligned(dO, 16) && skg_aligned(dQ, 8));
  AttnParams p{};
  p.Q = (const half_t*)Q; p.ldq = ldq; p.K = (const half_t*)K; p.ldk = ldk; p.V = (const half_t*)V; p.ldv = ldv;
  p.dO = (const half_t*)dO; p.lddo = lddo;
  p.lse = const_cast<float*>(lse); p.delta = delta; p.O = (half_t*)dQ; p.ldo = lddq;
  p.Of = (const half_t*)O; p.ldof = ldo; p.delta_out = delta_out;
  p.batch = batch; p.heads = heads; p.Nq = Nq; p.Nkv = Nkv; p.kv_stride = kv_stride; p.dh = dh; p.scale = scale;
  hipStream_t st = (hipStream_t)stream;
  p.nx = skg_cdiv(Nq, dh <= 40 ? 128 : 64);         // query tiles per wave: 2 up to d = 40, 1 beyond
  dim3 grid((unsigned)p.nx * heads * batch);
  switch (dh) {
    case 16: hipLaunchKernelGGL((attn_bwd_dq_kernel<1, 1, 2>), grid, dim3(256), 0, st, p); break;
    case 32: hipLaunchKernelGGL((attn_bwd_dq_kernel<1, 2, 2>), grid, dim3(256), 0, st, p); break;
    case 40: hipLaunchKernelGGL((attn_bwd_dq_kernel<2, 3, 2>), grid, dim3(256), 0, st, p); break;
    case 64: hipLaunchKernelGGL((attn_bwd_dq_kernel<2, 4, 1>), grid, dim3(256), 0, st, p); break;
    case 80: hipLaunchKernelGGL((attn_bwd_dq_kernel<3, 5, 1>), grid, dim3(256), 0, st, p); break;
    case 160: hipLaunchKernelGGL((attn_bwd_dq_kernel<5, 10, 1>), grid, dim3(256), 0, st, p); break;
    default: return SKG_E_UNSUPPORTED;
  }
  SKG_CHECK_LAUNCH("skg_attn_bwd_dq");
  return SKG_OK;
}

extern "C" int skg_attn_bwd_dq(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv,
                               const void* dO, int lddo, const float* lse,
                               const float* delta, void* dQ, int lddq, int batch, int heads, int Nq, int Nkv,
                               int kv_stride, int dh, float scale, void* stream) {
  SKG_REQUIRE(delta);
  return attn_bwd_dq_impl(Q, ldq, K, ldk, V, ldv, dO, lddo, lse, delta, dQ, lddq, batch, heads, Nq, Nkv, kv_stride, dh, scale, stream,
                          nullptr, 0, nullptr);
}

// ... with skg_attn_bwd_delta in its prologue (round 5): delta[b][h][q] = sum_d dO O is formed from the dO fragments the launch loads
// anyway, used, and stored to delta_out for the dK / dV launch that follows - one launch and one read of dO fewer per attention
extern "C" int skg_attn_bwd_dq_delta(const void* Q, int ldq, const void* K, int ldk, const void* V, int ldv, const void* dO, int lddo,
                                     const void* O, int ldo, const float* lse, float* delta_out, void* dQ, int lddq, int batch,
                                     int heads, int Nq, int Nkv, int kv_stride, int dh, float scale, void* stream) {
  SKG_REQUIRE(O && delta_out);
  return attn_bwd_dq_impl(Q, ldq, K, ldk, V, ldv, dO, lddo, lse, nullptr, dQ, lddq, batch, heads, Nq, Nkv, kv_stride, dh, scale, stream,
                          O, ldo, delta_out);
}

extern "C" int skg_attn_bwd_dkv(const void* Q, int ldq, const void* K, int ldk,
                                const void* V, int ldv, const void* dO, int lddo,
                                const float* lse, const float* delta, void* dK, int lddk, void* dV, int lddv,
                                int batch, int heads, int Nq, int Nkv, int dh, float scale, void* stream) {
  SKG_REQUIRE(Q && K && V && dO && lse && delta && dK && dV && common_ok(batch, heads, Nq, Nkv, Nkv, dh));
  SKG_REQUIRE(ldq % 8 == 0 && ldk % 8 == 0 && ldv % 8 == 0 && lddo % 8 == 0 && lddk % 4 == 0 && lddv % 4 == 0);
  SKG_REQUIRE(skg_aligned(Q, 16) && skg_aligned(K, 16) && skg_aligned(V, 16) &&
              skg_aligned(dO, 16) && skg_aligned(dK, 8) && skg_aligned(dV, 8));
  AttnParams p{};
  p.Q = (const half_t*)Q; p.ldq = ldq; p.K = (const half_t*)K; p.ldk = ldk;
  p.V = (const half_t*)V; p.ldv = ldv; p.dO = (const half_t*)dO; p.lddo = lddo;
  p.lse = const_cast<float*>(lse); p.delta = delta; p.O = (half_t*)dK; p.ldo = lddk;
  p.O2 = (half_t*)dV; p.ldo2 = lddv;
  p.batch = batch; p.heads = heads; p.Nq = Nq; p.Nkv = Nkv; p.kv_stride = Nkv; p.dh = dh; p.scale = scale;
  hipStream_t st = (hipStream_t)stream;
  p.nx = skg_cdiv(Nkv, dh <= 40 ? 128 : 64);        // key tiles per wave: 2 up to d = 40, 1 beyond (register budget)
  dim3 grid((unsigned)p.nx * heads * batch);
  switch (dh) {
    case 16: hipLaunchKernelGGL((attn_bwd_dkv_kernel<1, 1, 2>), grid, dim3(256), 0, st, p); break;
    case 32: hipLaunchKernelGGL((attn_bwd_dkv_kernel<1, 2, 2>), grid, dim3(256), 0, st, p); break;
    case 40: hipLaunchKernelGGL((attn_bwd_dkv_kernel<2, 3, 2>), grid, dim3(256), 0, st, p); break;
    case 64: hipLaunchKernelGGL((attn_bwd_dkv_kernel<2, 4, 1>), grid, dim3(256), 0, st, p); break;
    case 80: hipLaunchKernelGGL((attn_bwd_dkv_kernel<3, 5, 1>), grid, dim3(256), 0, st, p); break;
    case 160: hipLaunchKernelGGL((attn_bwd_dkv_kernel<5, 10, 1>), grid, dim3(256), 0, st, p); break;
    default: return SKG_E_UNSUPPORTED;
  }
  SKG_CHECK_LAUNCH("skg_attn_bwd_dkv");
  return SKG_OK;
}

#ifdef SKG_PHASES
extern "C" int skg_debug_attn_phases(void* host_out, int nwaves) {      // not part of the ABI: profiling builds only
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn_phase), (size_t)nwaves * 8 * sizeof(unsigned long long));
}
#endif

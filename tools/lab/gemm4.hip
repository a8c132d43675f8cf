// v4 fp16 MFMA GEMM for gfx950: persistent, wave-specialised 128 x 160 tiles (DIRECT mode, plain epilogue).
//
// Why: phase stamps on the K = 320 layers show a v2 workgroup computing for ~16 % of its life - the
// rest is the first DMA round trip of every tile and the tile's 41 KB of stores, and because loads and stores of one
// wave retire through ONE in-order counter (vmcnt) a wave cannot wait for its next operand without also waiting for
// its own stores.  Here ONE workgroup per CU (8 waves) runs all of its tiles back to back:
//   waves 0-3  "compute": issue the LDS-DMA of a 3-stage ring that runs AHEAD across tile boundaries (two K steps in
//              flight, no per-tile prologue bubble), do the MFMAs, and hand the finished tile to LDS (fp16, bias and
//              alpha applied).  They never touch global memory except through the DMA, so `s_waitcnt vmcnt(9)` is an
//              exact "oldest stage has landed";
//   waves 4-7  "store": while tile j computes, drain tile j-1 from the LDS staging area: residual add, ReLU, 16-byte
//              (optionally non-temporal) stores, in K-step sized chunks.  Their vmcnt only ever covers their own loads
//              and stores.
// Only s_barrier exists on gfx950 (no named barriers), so both roles execute the SAME barrier sequence: KT step
// barriers + R (staging free) + H (staging full) per tile, plus one drain round at the end.
//
// Scope of this first version (everything else stays on gemm2.hip): MODE_DIRECT, K % 64 == 0, N % 160 == 0, fp16
// output with 16-byte aligned rows, no fused GEGLU / fp32 output / split-K.  With a residual the sum is rounded twice
// (tile -> fp16 in LDS, + residual -> fp16): <= 1 ulp instead of 0.5.
#include "gemm_params.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 160, BK = 64, NSTG = 3;
constexpr int STAGE = (BM + BN) * BK;        // halves per ring stage: A tile then B tile (36 864 B)
constexpr int SP = BN + 8;                   // staging pitch in halves (336 B rows)
constexpr int NTHR = 512;
constexpr int MT = 4, NT = 5;                // compute wave tile 64 x 80
constexpr int ACH = 4, BCH = 5, ND = ACH + BCH;
constexpr unsigned OOB = 0x80000000u;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, half_t* lds_wave_base, unsigned voff,
                                      unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// issue-order plan of one K step (same as gemm2.hip): fragment reads >= one MFMA group ahead, DMA woven in between
template <int MASK, int N>
__device__ __forceinline__ void sgb() {
  if constexpr (N > 0) __builtin_amdgcn_sched_group_barrier(MASK, N, 0);
}
constexpr int SGB_MFMA = 0x008, SGB_VMEM = 0x010, SGB_DSREAD = 0x100;
template <int N, int R, int V>
__device__ __forceinline__ void sched_group() {
  if constexpr (N > 0) {
    sgb<SGB_MFMA, 1>();
    if constexpr (R > 0) {
      sgb<SGB_DSREAD, 1>();
      sched_group<N - 1, R - 1, V>();
    } else if constexpr (V > 0) {
      sgb<SGB_VMEM, 1>();
      sched_group<N - 1, 0, V - 1>();
    } else {
      sgb<SGB_MFMA, N - 1>();
    }
  }
}
constexpr int plan_reads(int g) {
  const int left = 2 * (MT + NT) - (NT + 1) - g * 3;
  return left < 0 ? 0 : (left < 3 ? left : 3);
}
constexpr int plan_vmem(int g) {          // compute waves fetch the A operand only: ACH instructions per step
  int left = ACH;
  for (int h = 0; h <= g; ++h) {
    const int v = (NT - plan_reads(h)) < left ? (NT - plan_reads(h)) : left;
    if (h == g) return v;
    left -= v;
  }
  return 0;
}
template <int G>
__device__ __forceinline__ void sched_groups() {
  if constexpr (G < 2 * MT) {
    sched_group<NT, plan_reads(G), plan_vmem(G)>();
    sched_groups<G + 1>();
  }
}

__global__ __launch_bounds__(NTHR, 1) void gemm4_kernel(const GemmParams p, int tiles_n, int ntiles, unsigned a_bytes,
                                                       unsigned b_bytes) {
  __shared__ __attribute__((aligned(16))) half_t smem[NSTG * STAGE + BM * SP + 2 * BN];
  half_t* const ring = smem;
  half_t* const stg = smem + NSTG * STAGE;
  float* const bias_s = reinterpret_cast<float*>(stg + BM * SP);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int KT = p.K / BK;
  const int nmine = (ntiles - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;      // tiles of this workgroup
  const bool relu = p.flags & SKG_EPI_RELU;
  const bool stream_out = p.flags & 0x800u;

  // tile k of this workgroup -> (m0, n0): XCD-aware order as in gemm2.hip (the grid is a multiple of 8, so a
  // workgroup keeps its XCD), column tiles of one row panel adjacent
  auto tile_mn = [&](int k, int& m0, int& n0) {
    const int vb = (int)blockIdx.x + k * (int)gridDim.x;
    const int xcd = vb & 7, idx = vb >> 3;
    const int q = ntiles >> 3, r = ntiles & 7;
    const int lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    const int tile_m = lid / tiles_n;
    m0 = tile_m * BM;
    n0 = (lid - tile_m * tiles_n) * BN;
  };

  if (wave < 4) {
    // =================================================================================== compute waves
    const int wm = wave >> 1, wn = wave & 1;
    const int g = lane >> 4, l16 = lane & 15;
    const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
    const int lr = lane >> 3, lq = lane & 7;     // DMA lane = (row of an 8-row chunk, 16-byte slot)

    // ---- issue side (A operand only; the store waves fetch B): runs two K steps ahead of the compute side
    unsigned a_voff[ACH];
    int ik = 0, ikt = 0, gi = 0;                  // next step to issue: tile ik, K tile ikt, global index gi
    auto set_issue_tile = [&](int k) {
      int m0, n0;
      tile_mn(k, m0, n0);
#pragma unroll
      for (int j = 0; j < ACH; ++j) {
        const int r = (j * 4 + wave) * 8 + lr;
        const unsigned pk = (unsigned)(lq ^ ((r >> 1) & 7)) * 16u;
        const int m = m0 + r;
        a_voff[j] = m < p.M ? (unsigned)m * (unsigned)p.lda * 2u + pk : OOB;
      }
    };
    struct DmaStep { unsigned va[ACH], so; int stage; };
    // always ACH instructions per step (out of range = zero fill once the stream has ended): vmcnt arithmetic is fixed
    auto next_dma = [&]() {
      DmaStep d;
      const bool live = ik < nmine;
#pragma unroll
      for (int j = 0; j < ACH; ++j) d.va[j] = live ? a_voff[j] : OOB;
      d.so = (unsigned)(ikt * BK) * 2u;
      d.stage = gi % NSTG;
      ++gi;
      if (live && ++ikt == KT) {
        ikt = 0;
        if (++ik < nmine) set_issue_tile(ik);
      }
      return d;
    };
    auto dma_one = [&](const DmaStep& d, int i) {
      dma16(rA, ring + d.stage * STAGE + (i * 4 + wave) * 8 * BK, d.va[i], d.so);
    };

    // ---- compute side
    int a_ad[MT][2], b_ad[NT][2];                 // fragment offsets (halves) inside a stage
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int row = wm * 64 + i * 16 + l16;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) a_ad[i][ks] = row * BK + (((ks * 4 + g) ^ ((row >> 1) & 7)) << 3);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int row = wn * 80 + j * 16 + l16;
#pragma unroll
      for (int ks = 0; ks < 2; ++ks) b_ad[j][ks] = BM * BK + row * BK + (((ks * 4 + g) ^ ((row >> 1) & 7)) << 3);
    }
    float4_t acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};

    // one K step: the A-DMA of step (current + 2) + 40 MFMAs on the current stage, one basic block, pinned order
    auto step = [&](int cstage) {
      const DmaStep d = next_dma();
      const half_t* sb = ring + cstage * STAGE;
      half8_t xf[2][MT], wf[2][NT];
      auto rd = [&](int idx) {
        if (idx < NT) wf[0][idx] = ld_half8(sb + b_ad[idx][0]);
        else if (idx < NT + MT) xf[0][idx - NT] = ld_half8(sb + a_ad[idx - NT][0]);
        else if (idx < 2 * NT + MT) wf[1][idx - NT - MT] = ld_half8(sb + b_ad[idx - NT - MT][1]);
        else xf[1][idx - 2 * NT - MT] = ld_half8(sb + a_ad[idx - 2 * NT - MT][1]);
      };
      int ri = 0, di = 0;
#pragma unroll
      for (int r = 0; r < NT + 1; ++r) rd(ri++);
#pragma unroll
      for (int gq = 0; gq < 2 * MT; ++gq) {
#pragma unroll
        for (int r = 0; r < plan_reads(gq); ++r) rd(ri++);
#pragma unroll
        for (int v = 0; v < plan_vmem(gq); ++v) dma_one(d, di++);
      }
#pragma unroll
      for (int ks = 0; ks < 2; ++ks)
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks][j], xf[ks][i], acc[i][j], 0, 0, 0);
      sgb<SGB_DSREAD, NT + 1>();
      sched_groups<0>();
    };

    // prologue: two steps in flight
    set_issue_tile(0);
#pragma unroll 1
    for (int s = 0; s < 2; ++s) {
      const DmaStep d = next_dma();
#pragma unroll
      for (int i = 0; i < ACH; ++i) dma_one(d, i);
    }
    int gc = 0;                                   // global index of the step being computed
    for (int k = 0; k < nmine; ++k) {
      for (int kt = 0; kt < KT; ++kt) {
        asm volatile("s_waitcnt vmcnt(4)" ::: "memory");      // A of step gc landed; step gc + 1 may still be in flight
        lds_barrier();                                        // ... A and B, for every wave; stage (gc + 2) % 3 is free
        step(gc % NSTG);
        ++gc;
      }
      lds_barrier();                                          // R: the store waves are done with the staging area
      // tile -> LDS: fp16((acc + bias) * alpha); lane holds C[row = .. + l16][4 consecutive columns]
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int r = wm * 64 + i * 16 + l16, c = wn * 80 + j * 16 + g * 4;
          const float4_t b = *reinterpret_cast<const float4_t*>(&bias_s[c]);
          const float4_t v = (acc[i][j] + b) * p.alpha;
          half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
          st_half4(stg + r * SP + c, o);
          acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
        }
      lds_barrier();                                          // H: staging holds tile k
    }
    // drain round: the store waves still have the last tile to write; keep the barrier count equal
    for (int kt = 0; kt < KT + 2; ++kt) lds_barrier();
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // (the trailing zero-fill DMAs)
  } else {
    // =================================================================================== store waves
    // Per K step, in this order: BCH weight-tile DMA instructions for step (current + 2), then `iters` pieces of the
    // previous tile: [residual load,] store.  Every step issues the SAME number of vector-memory instructions (out
    // of range = dropped by the buffer descriptor), so "the weight tile of the current step has landed" is the
    // immediate s_waitcnt vmcnt(BCH + 2 * per-step piece instructions) no matter how far the stores are behind.
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    const int sw = wave - 4;
    const int st = tid - 256;
    const int lr = lane >> 3, lq = lane & 7;
    const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, b_bytes, 0x00020000);
    const unsigned c_bytes = (unsigned)((((size_t)p.M - 1) * p.ldc + p.N) * 2);
    const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, c_bytes, 0x00020000);
    const unsigned r_bytes = p.res ? (unsigned)((((size_t)p.M - 1) * p.ldr + p.N) * 2) : 0u;
    const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)(p.res ? p.res : p.B), 0, r_bytes, 0x00020000);
    const bool has_res = p.res != nullptr;
    constexpr int PIECES = BM * (BN / 8);                     // 16-byte pieces of a tile (2560)
    const int iters = (KT >= 10) ? 1 : 2;                     // pieces per thread and K step (KT >= 5: launcher)
    const int per = iters * 256;

    unsigned b_voff[BCH];
    int ik = 0, ikt = 0, gi = 0;
    auto set_issue_tile = [&](int k) {
      int m0, n0;
      tile_mn(k, m0, n0);
#pragma unroll
      for (int j = 0; j < BCH; ++j) {
        const int r = (j * 4 + sw) * 8 + lr;
        const unsigned pk = (unsigned)(lq ^ ((r >> 1) & 7)) * 16u;
        const int n = n0 + r;
        b_voff[j] = n < p.N ? (unsigned)n * (unsigned)p.ldb * 2u + pk : OOB;
      }
    };
    auto issue_b = [&]() {
      const bool live = ik < nmine;
      half_t* base = ring + (gi % NSTG) * STAGE + BM * BK;
      const unsigned so = (unsigned)(ikt * BK) * 2u;
#pragma unroll
      for (int j = 0; j < BCH; ++j) dma16(rB, base + (j * 4 + sw) * 8 * BK, live ? b_voff[j] : OOB, so);
      ++gi;
      if (live && ++ikt == KT) {
        ikt = 0;
        if (++ik < nmine) set_issue_tile(ik);
      }
    };
    // pieces [kt * per, (kt + 1) * per) of the tile in the staging area -> global memory.  The residual of a chunk is
    // loaded ONE STEP EARLIER (res_load) so that using it never needs the instructions issued after it to complete.
    struct ResRegs { half8_t v[2]; };
    auto piece = [&](int kt, int it, int tm0, int tn0, int& r, int& c, int& m, int& n) {
      const int pi = kt * per + it * 256 + st;
      const int pc = min(pi, PIECES - 1);
      r = pc / (BN / 8);
      c = (pc - r * (BN / 8)) * 8;
      m = tm0 + r;
      n = tn0 + c;
      return pi < PIECES && m < p.M;
    };
    auto res_load = [&](bool real, int kt, int tm0, int tn0) {
      ResRegs rr;
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        rr.v[it] = zero_half8();
        if (it < iters && has_res) {
          int r, c, m, n;
          const bool ok = piece(kt, it, tm0, tn0, r, c, m, n) && real;
          const unsigned ro = ok ? (unsigned)(((size_t)m * p.ldr + n) * 2) : OOB;
          rr.v[it] = __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rR, ro, 0, 0));
        }
      }
      return rr;
    };
    auto chunk = [&](bool real, int kt, int tm0, int tn0, const ResRegs& rr) {
#pragma unroll
      for (int it = 0; it < 2; ++it) {
        if (it >= iters) break;
        int r, c, m, n;
        const bool ok = piece(kt, it, tm0, tn0, r, c, m, n) && real;
        half8_t v = ld_half8(stg + r * SP + c);
        if (has_res) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (half_t)((float)v[e] + (float)rr.v[it][e]);
        }
        if (relu) {
#pragma unroll
          for (int e = 0; e < 8; ++e) v[e] = (float)v[e] > 0.f ? v[e] : (half_t)0.f;
        }
        const unsigned co = ok ? (unsigned)(((size_t)m * p.ldc + n) * 2) : OOB;
        if (stream_out) __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rC, co, 0, 2);   // nt
        else __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rC, co, 0, 0);
      }
    };
    auto wait_b = [&]() {            // BCH + 2 * iters * (1 + has_res) younger instructions may stay in flight
      const int n = BCH + 2 * iters * (has_res ? 2 : 1);
      switch (n) {
        case 7: asm volatile("s_waitcnt vmcnt(7)" ::: "memory"); break;
        case 9: asm volatile("s_waitcnt vmcnt(9)" ::: "memory"); break;
        case 13: asm volatile("s_waitcnt vmcnt(13)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
      }
    };

    set_issue_tile(0);
    ResRegs rcur = res_load(false, 0, 0, 0);                  // prologue with the steady-state instruction pattern:
    issue_b();                                                //   [B-DMA x5] [residual loads] [stores] per step
    ResRegs rnew = res_load(false, 0, 0, 0);
    chunk(false, 0, 0, 0, rcur);
    rcur = rnew;
    issue_b();
    rnew = res_load(false, 0, 0, 0);
    chunk(false, 0, 0, 0, rcur);
    rcur = rnew;
    int pm0 = 0, pn0 = 0;                                     // tile whose data sits in the staging area
    for (int k = 0; k <= nmine; ++k) {
      int m0 = 0, n0 = 0;
      if (k < nmine) tile_mn(k, m0, n0);
      for (int kt = 0; kt < KT; ++kt) {
        wait_b();
        lds_barrier();
        if (kt == 0 && k < nmine && st < BN)
          bias_s[st] = (p.bias && n0 + st < p.N) ? (float)p.bias[n0 + st] : 0.f;      // read after R of this tile
        issue_b();
        // residual of the NEXT step's chunk: same staged tile, or (after the last step) the tile being computed now
        if (kt + 1 < KT) rnew = res_load(k > 0, kt + 1, pm0, pn0);
        else rnew = res_load(k < nmine, 0, m0, n0);
        chunk(k > 0, kt, pm0, pn0, rcur);
        rcur = rnew;
      }
      lds_barrier();      // R
      lds_barrier();      // H
      pm0 = m0; pn0 = n0;
    }
  }
}

}  // namespace

// Launches the persistent wave-specialised kernel if the call fits its scope; false = nothing launched.
bool skg_gemm4_try_launch(const GemmParams& p, int mode, hipStream_t st) {
  // Opt-in (SKG_GEMM4=1): on MI355X this kernel is correct (tests/_gemm4_check.py) but 5-25 % SLOWER than gemm2.hip -
  // measured 2 000-2 300 cycles per K step against 1 540 per tile-step for two co-resident v2 workgroups; see the
  // list of experiments in DESIGN.md 3b.  Kept as the starting point for the round-2 work on the short-K layers.
  static const bool on = getenv("SKG_GEMM4") != nullptr;
  if (!on || mode != MODE_DIRECT) return false;
  if (p.K % BK != 0 || p.K < 5 * BK || p.N % BN != 0 || p.M < 1) return false;
  if (p.flags & (SKG_EPI_OUT_F32 | SKG_EPI_GEGLU)) return false;
  if (p.ldc % 8 != 0 || (reinterpret_cast<uintptr_t>(p.C) & 15) != 0) return false;
  if (p.res && (p.ldr % 8 != 0 || (reinterpret_cast<uintptr_t>(p.res) & 15) != 0)) return false;
  const unsigned long long a = ((unsigned long long)(p.M - 1) * p.lda + p.K) * 2ull;
  const unsigned long long b = ((unsigned long long)(p.N - 1) * p.ldb + p.K) * 2ull;
  if (a >= 0x7fffffffull || b >= 0x7fffffffull) return false;
  if ((((unsigned long long)p.M - 1) * p.ldc + p.N) * 2ull >= 0x7fffffffull) return false;      // 32-bit store offsets
  if (p.res && (((unsigned long long)p.M - 1) * p.ldr + p.N) * 2ull >= 0x7fffffffull) return false;
  const int tiles_n = p.N / BN;
  const int ntiles = skg_cdiv(p.M, BM) * tiles_n;
  if (ntiles < 256) return false;                      // fewer tiles than CUs: gemm2 (+ split-K) keeps the chip busier
  GemmParams q = p;
  if ((size_t)p.M * p.N * 2 >= ((size_t)32 << 20)) q.flags |= 0x800u;          // streaming stores past the L2 size
  hipLaunchKernelGGL(gemm4_kernel, dim3(256), dim3(NTHR), 0, st, q, tiles_n, ntiles, (unsigned)a, (unsigned)b);
  return true;
}

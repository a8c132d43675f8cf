// v8 fp16 MFMA GEMM / 3x3 implicit-GEMM convolution for gfx950: ONE 8-wave workgroup per CU on a 256 x 160 tile,
// three LDS stages, and a PING-PONG schedule between the two waves that share a SIMD.
//
// Why: rocprofv3 counters put the matrix pipe of the v2 convolution kernel (two co-resident
// 4-wave workgroups on 128 x 160 tiles) at 44 % busy at 2.17 GHz.  In v2 every wave interleaves LDS fragment reads,
// LDS-DMA issue (each `buffer_load ... lds` occupies its wave for ~100 cycles) and MFMAs in one instruction stream,
// and the only thing that covers a wave's non-MFMA time is whatever the co-resident workgroup happens to be doing.
// Here the overlap is constructed instead of hoped for.  Waves w and w + 4 of a workgroup land on the same SIMD;
// group 0 (waves 0-3) and group 1 (waves 4-7) run the SAME sequence of segments
//        LOAD(j): 9 fragment reads of k-sub-step j + 3-4 LDS-DMA instructions of the tile two ahead -> s_barrier ->
//        COMPUTE(j): 20 MFMAs out of registers at raised priority                                     -> s_barrier
// but group 1 starts one barrier late, so on every SIMD one wave streams MFMAs while its partner reads and issues
// DMA (the 8-phase idea of the CDNA4 GEMM playbook, cut to the two phases per 32-deep sub-step this tile needs).
// A 256 x 160 tile moves 10 B of operands per kFLOP through the texture path instead of 14 (128 x 160).
//
// Pipeline bookkeeping (tile t = 64-deep K step, sub-step j = 2t + ks, stage = t % 3):
//   * tile t + 2 is requested during LOAD(2t) (A rows) and LOAD(2t + 1) (B rows) into stage (t + 2) % 3 = (t - 1) % 3,
//     whose last reader - group 1's LOAD(2t - 1) - finished two barriers earlier;
//   * every wave issues exactly ND DMA instructions per tile (7 for waves 0-3, 6 for waves 4-7: 20 B chunks over 8
//     waves; out-of-range ones are zero fills), so `s_waitcnt vmcnt(ND)` at the end of LOAD(2t + 1) means "tile t + 1
//     has landed" for that wave, and the barrier that follows publishes it to the others before anyone reads it.
// Two tiles: 256 x 160 (wave grid 4 x 2, three stages, prefetch distance two tiles) and 256 x 320 (wave grid 2 x 4, wave
// tile 128 x 80, TWO stages of 73.7 KB: the whole next tile is requested in LOAD(2t) and waited for with vmcnt(0) at the
// end of LOAD(2t + 1); 7 B of operands per kFLOP) - the latter only where one workgroup per CU still fills the chip,
// i.e. the 64 x 64 level at 16 rows.
// Scope: MODE_DIRECT and MODE_S1, K % 64 == 0 (conv: Cin % 64 == 0), N % 160 == 0, fp16 output, no fused GEGLU, no
// split-K; everything else stays on gemm2.hip.  Epilogue: bias / alpha / residual / ReLU in fp32, one rounding.
#include "gemm_params.h"
#include <stdlib.h>

namespace {

constexpr int BM = 256, BK = 64, NW = 8, NTHR = 512;
constexpr int NT = 5, ACH = 4;                   // every wave tile is 80 columns wide; 32 A chunks of 8 rows over 8 waves
constexpr unsigned OOB = 0x80000000u;

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, half_t* lds_wave_base, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void bar() { asm volatile("s_barrier" ::: "memory"); }

// EXP: probe builds (SKG_G8_EXP; compute on stale / missing data, timing only): 1 = no DMA in the loop, 2 = no
// fragment reads in the loop, 4 = no priority raise, 8 = no stagger between the two groups
// HILO: the accuracy-mode epilogue (skg_*_hilo): the residual is the pair p.res + p.res_lo, the output the pair
// hi = fp16(v) -> p.C, lo = fp16(v - hi) -> p.c_lo (same leading dimensions); statistics (GNS) are those of hi
template <int BN, int MODE, int EXP = 0, bool GNS = false, bool HILO = false>
__global__ __launch_bounds__(NTHR, 2) void gemm8_kernel(const GemmParams p, int tiles_n, int nwg, unsigned a_bytes,
                                                        unsigned b_bytes, unsigned a_shift) {
  constexpr int NS = BN == 160 ? 3 : 2;            // LDS stages
  constexpr int WGN = BN / 80, WGM = NW / WGN;     // wave grid: 4 x 2 (BN = 160) or 2 x 4 (BN = 320)
  constexpr int WM = BM / WGM, WN = 80, MT = WM / 16;
  constexpr int BCH = (BN / 8 + NW - 1) / NW;      // B chunks of 8 rows per wave: 3 (20 chunks: the last slot is idle
                                                   // for waves 4-7) or 5 (40 chunks)
  constexpr int ND = ACH + BCH;
  constexpr int STAGE = (BM + BN) * BK;            // halves per stage: A tile then B tile
  constexpr bool SINK = (BN / 8) % NW != 0;
  __shared__ __attribute__((aligned(16))) half_t smem[NS * STAGE + (SINK ? 512 : 0)];   // + 1 KB sink for an idle DMA slot
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                       // waves w and w + 4 share a SIMD: one of each group
  const int wm = WGN == 2 ? (wave & 3) : (wave >> 2), wn = WGN == 2 ? (wave >> 2) : (wave & 3);
  const int g = lane >> 4, l16 = lane & 15;

  const __amdgpu_buffer_rsrc_t rA =
      __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A - a_shift), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, b_bytes, 0x00020000);
  // second A operand (the ResnetBlock's 1x1 shortcut folded in as K tiles past the 3x3 walk: gemm_params.h): plain rows, no halo
  const __amdgpu_buffer_rsrc_t rA2 = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(p.A2 ? p.A2 : p.A), 0, p.A2 ? ((unsigned)(p.M - 1) * (unsigned)p.lda2 + (unsigned)p.K2) * 2u : 0u, 0x00020000);
  const int KT1 = (MODE == MODE_S1 && p.K2) ? (p.K - p.K2) / BK : (1 << 30);

  // XCD-aware tile assignment (workgroup b -> XCD b % 8): every XCD owns a contiguous tile range, n fastest
  int lid;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int tile_m = lid / tiles_n, tile_n = lid - tile_m * tiles_n;
  const int m0 = tile_m * BM, n0 = tile_n * BN;

  // ---- per-lane DMA source description (as gemm2.hip: one voffset per operand row, tap / k position in soffset) ----
  const int lr = lane >> 3, lq = lane & 7;
  unsigned a_voff[ACH], a_mask[ACH];
#pragma unroll
  for (int j = 0; j < ACH; ++j) {
    const int r = (j * NW + wave) * 8 + lr;
    const unsigned pk = (unsigned)(lq ^ ((r >> 1) & 7)) * 16u;
    const int m = m0 + r;
    const bool ok = m < p.M;
    const int mm = ok ? m : 0;
    a_mask[j] = 0;
    if (MODE == MODE_DIRECT) {
      a_voff[j] = ok ? (unsigned)mm * (unsigned)p.lda * 2u + pk : OOB;
    } else {
      const int ohw = p.OH * p.OW;
      const int b = mm / ohw;
      const int rr = mm - b * ohw;
      const int oy = rr / p.OW, ox = rr - oy * p.OW;
      const unsigned img = (unsigned)b * (unsigned)(p.IH * p.IW);
      a_voff[j] = ok ? ((img + (unsigned)(oy * p.IW + ox)) * (unsigned)p.lda) * 2u + pk : OOB;
      unsigned mk = 0;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int iy = oy + t / 3 - 1, ix = ox + t % 3 - 1;
        if (ok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW) mk |= 1u << t;
      }
      a_mask[j] = mk;
    }
  }
  unsigned b_voff[BCH];
#pragma unroll
  for (int j = 0; j < BCH; ++j) {
    const int r = (j * NW + wave) * 8 + lr;
    const unsigned pk = (unsigned)(lq ^ ((r >> 1) & 7)) * 16u;
    const int n = n0 + r;
    b_voff[j] = (r < BN && n < p.N) ? (unsigned)n * (unsigned)p.ldb * 2u + pk : OOB;
  }
  const bool last_b_idle = SINK && wave >= (BN / 8) % NW;      // this wave's last B slot has no chunk: it zero-fills the sink

  const int KT = p.K / BK;
  // scalar description of K tile kt: A soffset, B soffset, filter tap (conv: channel block outermost, taps innermost)
  auto ktile = [&](int kt, unsigned& soa, unsigned& sob, int& tap) {
    if (MODE == MODE_DIRECT) {
      soa = sob = (unsigned)kt * (BK * 2u);
      tap = 0;
    } else {
      const int cb = kt / 9;
      tap = kt - cb * 9;
      const int c0 = cb * BK;
      const int ky = tap / 3, kx = tap - ky * 3;
      soa = (unsigned)((ky * p.IW + kx) * p.lda + c0) * 2u;
      sob = (unsigned)(tap * p.Cin + c0) * 2u;
    }
  };
  auto dma_a = [&](int kt, int buf) {        // the tile's A rows: ACH instructions
    unsigned soa, sob;
    int tap;
    const bool live = kt < KT;
    if (MODE == MODE_S1 && live && kt >= KT1) {      // (wave-uniform) the shortcut operand: row m of A2, channels (kt - KT1) * 64 ..
#pragma unroll
      for (int j = 0; j < ACH; ++j) {
        const int r = (j * NW + wave) * 8 + lr;
        const unsigned pk = (unsigned)(lq ^ ((r >> 1) & 7)) * 16u;
        dma16(rA2, &smem[buf * STAGE + (j * NW + wave) * 8 * BK], m0 + r < p.M ? (unsigned)(m0 + r) * (unsigned)p.lda2 * 2u + pk : OOB,
              (unsigned)(kt - KT1) * (BK * 2u));
      }
      return;
    }
    ktile(live ? kt : 0, soa, sob, tap);
#pragma unroll
    for (int j = 0; j < ACH; ++j) {
      unsigned v = a_voff[j];
      if (MODE != MODE_DIRECT) v = ((a_mask[j] >> tap) & 1u) ? v : OOB;
      dma16(rA, &smem[buf * STAGE + (j * NW + wave) * 8 * BK], live ? v : OOB, soa);
    }
  };
  auto dma_b = [&](int kt, int buf) {        // the tile's B rows: BCH instructions (an idle slot zero-fills the sink)
    unsigned soa, sob;
    int tap;
    const bool live = kt < KT;
    ktile(live ? kt : 0, soa, sob, tap);
    if (MODE == MODE_S1 && live && kt >= KT1) sob = (unsigned)kt * (BK * 2u);      // weight row = [9 taps x Cin | K2]
#pragma unroll
    for (int j = 0; j < BCH; ++j) {
      half_t* dst = (j == BCH - 1 && last_b_idle) ? &smem[NS * STAGE] : &smem[buf * STAGE + BM * BK + (j * NW + wave) * 8 * BK];
      dma16(rB, dst, live ? b_voff[j] : OOB, sob);
    }
  };

  float4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};

  // fragment read addresses (halves) inside a stage for k-sub-step 0 / 1; slot = piece ^ ((row >> 1) & 7).  Every wave
  // tile starts at a multiple of 16 rows and fragment i starts 16 i rows further, so the swizzle key is (l16 >> 1) & 7
  // for all of them: ONE address per operand and sub-step, the fragment index is an immediate offset.
  const int key = (l16 >> 1) & 7;
  int a_base[2], b_base[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_base[ks] = (wm * WM + l16) * BK + (((ks * 4 + g) ^ key) << 3);
    b_base[ks] = BM * BK + (wn * WN + l16) * BK + (((ks * 4 + g) ^ key) << 3);
  }

  half8_t xf[MT], wf[NT];
  // one sub-step of one wave: LOAD segment, barrier, COMPUTE segment, barrier
  auto substep = [&](int t, int stg, int ks) {        // stg, ks: compile-time constants at every call site
    const half_t* sb = &smem[stg * STAGE];
    if (!(EXP & 2) || t == 0) {
#pragma unroll
      for (int j = 0; j < NT; ++j) wf[j] = ld_half8(sb + b_base[ks] + j * 16 * BK);
#pragma unroll
      for (int i = 0; i < MT; ++i) xf[i] = ld_half8(sb + a_base[ks] + i * 16 * BK);
    }
    if (!(EXP & 1)) {
      if (NS == 3) {          // tile t + 2 into stage (stg + 2) % 3, whose last reader finished two barriers ago
        const int nbuf = stg == 0 ? 2 : stg - 1;
        if (ks == 0) dma_a(t + 2, nbuf);
        else dma_b(t + 2, nbuf);
      } else if (ks == 0) {   // tile t + 1 into the other stage: its last reader was group 1's LOAD(2t - 1), one barrier ago
        dma_a(t + 1, stg ^ 1);
        dma_b(t + 1, stg ^ 1);
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (ks == 1 && !(EXP & 1)) {
      // the tile that LOAD(2t + 2) reads must have landed before the barrier below.  Three stages: the newest tile's
      // ND instructions may stay in flight.  Two stages: the tile requested in LOAD(2t) is the one - drain.
      if (NS == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND) : "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    bar();
    __builtin_amdgcn_sched_barrier(0);
    if (!(EXP & 4)) __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
    if (!(EXP & 4)) __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    bar();
    __builtin_amdgcn_sched_barrier(0);
  };

  // prologue: the first NS - 1 tiles in flight, tile 0 landed and published
  dma_a(0, 0); dma_b(0, 0);
  if (NS == 3) {
    dma_a(1, 1); dma_b(1, 1);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(ND) : "memory");
  } else {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  }
  bar();
  if (EXP & 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  if (grp == 1 && !(EXP & 8)) bar();                   // group 1 runs one barrier behind group 0
  for (int t = 0; t < KT; t += NS) {
    substep(t, 0, 0);
    substep(t, 0, 1);
    if (t + 1 < KT) {
      substep(t + 1, 1, 0);
      substep(t + 1, 1, 1);
    }
    if (NS == 3 && t + 2 < KT) {
      substep(t + 2, 2, 0);
      substep(t + 2, 2, 1);
    }
  }
  if (grp == 0 && !(EXP & 8)) bar();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the last tiles' zero-fill DMA

  // ---- epilogue: lane holds C[m = .. + l16][n = .. + 4g .. 4g + 3] --------------------------------------------------
  // Every load is issued before the first use (conditions hoisted out of the element loops: a per-element
  // "load or not" select makes hipcc wait vmcnt(0) after each load - 40 serial round trips); four 16-row groups at a
  // time so that the residual registers stay within what the dead fragment registers free.  Rows beyond M read row
  // M - 1 and are not stored; N is a multiple of the tile width.
  const bool relu = p.flags & SKG_EPI_RELU;
  float4_t bv[NT];
#pragma unroll
  for (int j = 0; j < NT; ++j) bv[j] = float4_t{0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const half4_t b = ld_half4(p.bias + n0 + wn * WN + j * 16 + g * 4);
      bv[j] = float4_t{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
    }
  }
  const int mrow = m0 + wm * WM + l16;
  // GroupNorm statistics of the output (256 x 320 tile only; launcher-checked: whole tiles, WM = 128-row chunks inside
  // one sample, groups inside one wave's 80 columns): sum(y), sum(y^2) of the fp16 pairs a lane stores (v_dot2), summed
  // over the wave's 8 row blocks in registers, over the 16 row lanes by DPP adds, folded into groups through a
  // 640-byte per-wave LDS scratch - fixed order, no atomics.
  constexpr bool gn = GNS && BN == 320;      // own instantiation: the plain launches keep the lean epilogue
  __shared__ float gn_scr[gn ? NW * 80 : 1];
  float gs[NT][2][2];
#pragma unroll
  for (int j = 0; j < NT; ++j) gs[j][0][0] = gs[j][0][1] = gs[j][1][0] = gs[j][1][1] = 0.f;
  constexpr int RG = HILO ? 2 : 4;      // 16-row groups per pass (pair residual: twice the registers per group)
#pragma unroll
  for (int i0 = 0; i0 < MT; i0 += RG) {
    if (p.res) {      // (HILO: the launcher takes the launch only when p.res and p.res_lo are both there or both absent)
      half4_t rv[RG][NT], rl[RG][NT];
#pragma unroll
      for (int ii = 0; ii < RG; ++ii) {
        const int m = min(mrow + (i0 + ii) * 16, p.M - 1);
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const size_t off = (size_t)m * p.ldr + n0 + wn * WN + j * 16 + g * 4;
          rv[ii][j] = ld_half4(p.res + off);
          if (HILO) rl[ii][j] = ld_half4(p.res_lo + off);
        }
      }
#pragma unroll
      for (int ii = 0; ii < RG; ++ii)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const half4_t r = rv[ii][j];
          acc[i0 + ii][j] = (acc[i0 + ii][j] + bv[j]) * p.alpha + float4_t{(float)r[0], (float)r[1], (float)r[2], (float)r[3]};
          if (HILO) {
            const half4_t l = rl[ii][j];
            acc[i0 + ii][j] += float4_t{(float)l[0], (float)l[1], (float)l[2], (float)l[3]};
          }
        }
    } else {
#pragma unroll
      for (int ii = 0; ii < RG; ++ii)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i0 + ii][j] = (acc[i0 + ii][j] + bv[j]) * p.alpha;
    }
    if (relu) {
#pragma unroll
      for (int ii = 0; ii < RG; ++ii)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
          for (int e = 0; e < 4; ++e) acc[i0 + ii][j][e] = fmaxf(acc[i0 + ii][j][e], 0.f);
    }
#pragma unroll
    for (int ii = 0; ii < RG; ++ii) {
      const int m = mrow + (i0 + ii) * 16;
      if (m >= p.M) continue;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const float4_t v = acc[i0 + ii][j];
        const half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        const size_t off = (size_t)m * p.ldc + n0 + wn * WN + j * 16 + g * 4;
        st_half4(reinterpret_cast<half_t*>(p.C) + off, o);
        if (HILO) {
          const half4_t lo = {(half_t)(v[0] - (float)o[0]), (half_t)(v[1] - (float)o[1]), (half_t)(v[2] - (float)o[2]),
                              (half_t)(v[3] - (float)o[3])};
          st_half4(p.c_lo + off, lo);
        }
        if (gn) {
          typedef _Float16 h2 __attribute__((ext_vector_type(2)));
          const h2 one = {(_Float16)1.f, (_Float16)1.f}, lo = {o[0], o[1]}, hi = {o[2], o[3]};
          gs[j][0][0] = __builtin_amdgcn_fdot2(lo, one, gs[j][0][0], false);
          gs[j][0][1] = __builtin_amdgcn_fdot2(lo, lo, gs[j][0][1], false);
          gs[j][1][0] = __builtin_amdgcn_fdot2(hi, one, gs[j][1][0], false);
          gs[j][1][1] = __builtin_amdgcn_fdot2(hi, hi, gs[j][1][1], false);
        }
      }
    }
  }
  if constexpr (BN == 320) {
    if (gn) {
      float* const scr = gn_scr + wave * 80;       // [40 column pairs of the wave's 80 columns][sum, sum of squares]
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int h = 0; h < 2; ++h)
#pragma unroll
          for (int q = 0; q < 2; ++q) {
            float t = gs[j][h][q];
            t = dpp_add<0x128>(t); t = dpp_add<0x124>(t); t = dpp_add<0x4E>(t); t = dpp_add<0xB1>(t);     // over the 16 row lanes
            if (l16 == 0) scr[(j * 8 + g * 2 + h) * 2 + q] = t;
          }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // same wave wrote it: in order, visible after the wait
      const int cpg = p.N / p.gn_groups, ngr = WN / cpg;
      if (lane < 2 * ngr) {
        const int gi = lane >> 1, q = lane & 1;
        float t = 0.f;
        for (int c = 0; c < (cpg >> 1); ++c) t += scr[(gi * (cpg >> 1) + c) * 2 + q];
        const int row0 = m0 + wm * WM;
        const int nch = p.gn_hw >> 7, b = row0 / p.gn_hw, chunk = (row0 - b * p.gn_hw) >> 7;
        p.gn_partial[(((size_t)b * nch + chunk) * p.gn_groups + (n0 + wn * WN) / cpg + gi) * 2 + q] = t;
      }
    }
  }
}

inline bool operand_bytes(const GemmParams& p, int mode, unsigned long long& a, unsigned long long& b,
                          unsigned long long& shift) {
  b = ((unsigned long long)(p.N - 1) * p.ldb + p.K) * 2ull;
  if (mode == MODE_DIRECT) {
    shift = 0;
    a = ((unsigned long long)(p.M - 1) * p.lda + p.K) * 2ull;
  } else {
    const unsigned long long rows = (unsigned long long)p.M / ((unsigned long long)p.OH * p.OW);
    shift = (unsigned long long)(p.IW + 1) * p.lda * 2ull;
    a = rows * p.IH * p.IW * p.lda * 2ull + shift + (unsigned long long)(2 * p.IW + 2) * p.lda * 2ull;
  }
  return a < 0x7fffffffull && b < 0x7fffffffull;
}

// SKG_GEMM8 (read once): 0 = off; 1 = both tiles for every eligible shape (lab build only: `make lab`; the 256 x 160 form
// and the probe instantiations are not part of libskg.so); 2 = the 256 x 320 tile for
// long-K launches (K >= 1024) of both modes; unset = what ships: the 256 x 320 tile for 3x3 convolutions only - same
// box, interleaved (tools/gemm8_bench.py, profiles/r02_gemm8_320.txt): conv 640->320 @ 64x64 213-221 -> 199 us, 960->320
// 306-327 -> 275-292 us (1.24-1.32 PFLOP/s), 320->320 124-127 -> 121 us; the FF2 GEMM (K = 1280) is 5 % SLOWER with the
// direct-store epilogue and stays on gemm2.hip, like every shape the 256 x 160 tile would take.
int gemm8_mode() {
  static const int v = getenv("SKG_GEMM8") ? atoi(getenv("SKG_GEMM8")) : 3;
#ifndef SKG_LAB
  if (v == 1) return 3;
#endif
  return v;
}

// tile width v8 would use for this launch (0 = not taken)
int gemm8_tile(const GemmParams& p, int mode) {
  const int md = gemm8_mode();
  if (!md) return 0;
  if (mode != MODE_DIRECT && mode != MODE_S1) return 0;
  if (p.K % BK != 0 || p.K < 2 * BK || p.M < 1) return 0;
  if (mode == MODE_S1 && p.Cin % BK != 0) return 0;
  if (p.flags & (SKG_EPI_OUT_F32 | SKG_EPI_GEGLU)) return 0;
  const bool hilo = p.c_lo || p.res_lo;      // accuracy mode: the 256 x 320 tile has the hi / lo epilogue, the rest is gemm2.hip's
  if (hilo && (mode != MODE_S1 || !p.c_lo || (p.res != nullptr) != (p.res_lo != nullptr))) return 0;
  if (p.ntaps || p.up2 || p.seg_rows) return 0;        // polyphase upsample / segmented rows: gemm2.hip's tap walk / row map
  if (p.ldc % 4 != 0 || (p.res && p.ldr % 4 != 0)) return 0;
  unsigned long long a, b, s;
  if (!operand_bytes(p, mode, a, b, s)) return 0;
  const long tm = skg_cdiv(p.M, BM);
  // 256 x 320 where one workgroup per CU fills the chip in whole rounds; else 256 x 160 (mode 1 only)
  if (md == 3 && mode != MODE_S1) return 0;
  if (p.N % 320 == 0 && (md == 1 || p.K >= 1024)) {
    const long t = tm * (p.N / 320);
    if (t >= 224 && (t <= 256 || t >= 480)) return 320;
  }
  if (md == 1 && !hilo && p.N % 160 == 0 && tm * (p.N / 160) >= 224) return 160;
  return 0;
}

// the 256 x 320 instantiation writes the GroupNorm partials itself when every tile is whole, the 128-row halves stay
// inside one sample and no group straddles a wave's 80 columns
inline bool gn_fusable8(const GemmParams& p, int mode) {
#ifdef SKG_LAB
  // probe instantiations (SKG_G8_EXP) have no statistics epilogue: the caller must run the stand-alone pass
  static const bool probe = getenv("SKG_G8_EXP") && atoi(getenv("SKG_G8_EXP")) != 0;
  if (probe) return false;
#endif
  if (!p.gn_partial || p.gn_groups <= 0 || p.gn_hw <= 0 || mode != MODE_S1 || gemm8_tile(p, mode) != 320) return false;
  if (p.M % 256 != 0 || p.gn_hw % 128 != 0 || p.M % p.gn_hw != 0 || p.N % p.gn_groups != 0) return false;
  const int cpg = p.N / p.gn_groups;
  return !(cpg & 1) && 80 % cpg == 0;
}

template <int BN>
void launch8(const GemmParams& p_in, int mode, hipStream_t st) {
  GemmParams p = p_in;
  if (BN == 320 && gn_fusable8(p, mode)) p.flags |= SKG_FLAG_GN_STATS;
  unsigned long long a, b, s;
  operand_bytes(p, mode, a, b, s);
  const int tiles_n = p.N / BN;
  const int ntiles = skg_cdiv(p.M, BM) * tiles_n;
#ifdef SKG_LAB
  static const int exp = getenv("SKG_G8_EXP") ? atoi(getenv("SKG_G8_EXP")) : 0;
#else
  constexpr int exp = 0;
#endif
#define G8_LAUNCH(M_, E_) hipLaunchKernelGGL((gemm8_kernel<BN, M_, E_>), dim3(ntiles), dim3(NTHR), 0, st, p, tiles_n, ntiles, \
                                             (unsigned)a, (unsigned)b, (unsigned)s)
  if (mode == MODE_DIRECT) {
    G8_LAUNCH(MODE_DIRECT, 0);
  } else {
    switch (exp) {
#ifdef SKG_LAB
      case 1: G8_LAUNCH(MODE_S1, 1); break;
      case 2: G8_LAUNCH(MODE_S1, 2); break;
      case 3: G8_LAUNCH(MODE_S1, 3); break;
      case 4: G8_LAUNCH(MODE_S1, 4); break;
      case 8: G8_LAUNCH(MODE_S1, 8); break;
#endif
      default:
        if constexpr (BN == 320) {
          if (p.c_lo || p.res_lo) {      // accuracy mode
            if (p.flags & SKG_FLAG_GN_STATS)
              hipLaunchKernelGGL((gemm8_kernel<BN, MODE_S1, 0, true, true>), dim3(ntiles), dim3(NTHR), 0, st, p, tiles_n, ntiles,
                                 (unsigned)a, (unsigned)b, (unsigned)s);
            else
              hipLaunchKernelGGL((gemm8_kernel<BN, MODE_S1, 0, false, true>), dim3(ntiles), dim3(NTHR), 0, st, p, tiles_n, ntiles,
                                 (unsigned)a, (unsigned)b, (unsigned)s);
            break;
          }
          if (p.flags & SKG_FLAG_GN_STATS) {
            hipLaunchKernelGGL((gemm8_kernel<BN, MODE_S1, 0, true>), dim3(ntiles), dim3(NTHR), 0, st, p, tiles_n, ntiles,
                               (unsigned)a, (unsigned)b, (unsigned)s);
            break;
          }
        }
        G8_LAUNCH(MODE_S1, 0);
        break;
    }
  }
#undef G8_LAUNCH
}

}  // namespace

bool skg_gemm8_eligible(const GemmParams& p, int mode) { return gemm8_tile(p, mode) != 0; }
bool skg_gemm8_fuses_gn(const GemmParams& p, int mode) { return gn_fusable8(p, mode); }
int skg_gemm8_tile_n(const GemmParams& p, int mode) { return gemm8_tile(p, mode); }

bool skg_gemm8_try_launch(const GemmParams& p, int mode, hipStream_t st) {
  const int bn = gemm8_tile(p, mode);
  if (!bn) return false;
  if (bn == 320) launch8<320>(p, mode, st);
#ifdef SKG_LAB
  else launch8<160>(p, mode, st);
#else
  else return false;
#endif
  return true;
}

// v9 fp16 MFMA GEMM / 3x3 implicit-GEMM convolution for gfx950 with a HAND-PLACED K loop.
//
// What is different from gemm8.hip (two wave groups alternating LOAD / COMPUTE segments, four barriers per 64-deep K tile,
// all LDS-DMA of a tile issued in one burst) and from gemm2.hip (compiler-scheduled stage with sched_group_barrier hints):
//   * every wave runs ONE self-pipelined instruction stream: the K tile is cut into four 16-deep steps of 32x32x16 MFMAs;
//     while the MT x 5 MFMAs of step s issue out of fragment register set s & 1, the MT + 5 ds_read_b128 of step s + 1 go
//     into the other set - a register double buffer (28 VGPRs per set at MT = 2) that the 16x16x32 form of the same wave
//     tile cannot afford (56 per set).  A wave never waits for an LDS read it issued less than one step (>= 320 MFMA
//     cycles) ago;
//   * ONE s_barrier per K tile (inside step 2, after the tile's last fragment reads have returned and the wave's own DMA
//     of tile t + 1 has landed): it publishes tile t + 1 and frees tile t's stage at the same time;
//   * the 72 LDS-DMA instructions of a tile (256 x 320: 73.7 KB) are spread EVENLY over the MFMA slots of three (or four)
//     steps - one per ~3 MFMAs and wave - instead of 36 at once from four waves: the texture path takes one 1 KB
//     instruction per ~25 cycles and CU, a burst parks the issuing waves in front of it (gemm8's LOAD(2t) segment was
//     longer than the COMPUTE segment beside it for exactly that reason);
//   * issue order is pinned slot by slot (`__builtin_amdgcn_sched_barrier(0)` after every MFMA + its fillers): what the
//     ISA shows is what the source says; waits are hipcc's own exact lgkmcnt counts for the fragment reads (nothing to
//     wait for in steady state) and ONE hand-written `s_waitcnt vmcnt(0)` per tile in front of the barrier.
// Two waves per SIMD (8-wave workgroup, 256 x 320 tile, wave tile 64 x 160) run the same stream; whichever has an MFMA
// ready takes the matrix pipe, so a wave parked at the barrier or behind a DMA issue is covered by its partner.
//
// Geometry <WGM, MT>: wave grid WGM (M) x 2 (N), wave tile 32 MT x 160, block tile 32 MT WGM x 320, 64 WGM x 2 threads:
//   <4, 2> 256 x 320, 8 waves, <= 256 VGPRs  - the 64 x 64 level (256 tiles at 16 rows)
//   <2, 4> 256 x 320, 4 waves, <= 512 VGPRs  - one wave per SIMD, 147 KB instead of 229 KB of fragment reads per K tile
//   <2, 2> 128 x 320, 4 waves                - the 32 x 32 level (256 tiles at 16 rows, N = 640)
// LDS: [A stage 0 | A stage 1 | B stage 0 | B stage 1], rows of 64 halves, 16-byte piece p of row r at slot
// p ^ ((r >> 1) & 7) (gemm2.hip's swizzle: applied to the DMA source offset and to the read address).  A 32x32x16
// fragment read takes rows r0 + (lane & 31), piece 2 s + (lane >> 5): the 16 lanes of a ds_read_b128 service group sit
// in one half-wave, i.e. 16 different rows at one piece index = 8 keys x {even, odd row} = 16 distinct 16-byte slots of the
// 256-byte bank row: conflict-free.
// Scope: MODE_DIRECT / MODE_S1, K % 64 == 0, N % 320 == 0, fp16 output, bias / alpha / residual / ReLU epilogue; everything
// else stays where it was.
#include "gemm_params.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int BK = 64, BN = 320, NT = 5;
constexpr unsigned OOB = 0x80000000u;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef float float16_t __attribute__((ext_vector_type(16)));
template <int I>
using ic = std::integral_constant<int, I>;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, half_t* lds_wave_base, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void bar() { asm volatile("s_barrier" ::: "memory"); }
#define G9_PIN() __builtin_amdgcn_sched_barrier(0)

// compile-time loop: f(ic<0>{}), ..., f(ic<N - 1>{})
template <int N, int I = 0, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(ic<I>{});
    static_for<N, I + 1>(f);
  }
}

// VAR: bits 0-1 = DMA window of a tile, in steps after the barrier: 0 = three (even), 1 = two, 2 = one (front-loaded: more
// time to land before the next barrier's vmcnt(0)), 3 = four (the first half of step 2 as well); 4 = the second wave group
// sleeps ~128 cycles after every barrier (the two waves of a SIMD half a step apart: their read / DMA bursts interleave);
// 32 = barrier in the middle of step 2 with that step's fragment reads two per slot; 64 = fragment reads two per slot in
// every step; probes (wrong results, timing only): 8 = no DMA in the loop, 16 = no fragment reads
template <int WGM, int MT, int MODE, int VAR>
__global__ __launch_bounds__(WGM * 128, WGM == 4 ? 2 : 1) void gemm9_kernel(const GemmParams p, int tiles_n, int nwg,
                                                                            unsigned a_bytes, unsigned b_bytes,
                                                                            unsigned a_shift) {
  constexpr int NW = WGM * 2, BM = WGM * MT * 32;
  constexpr int ACH = (BM / 8) / NW, BCH = (BN / 8) / NW, ND = ACH + BCH;      // 1 KB DMA chunks (8 rows) per wave and tile
  constexpr int NM = MT * NT, NR = MT + NT;                                    // MFMAs / fragment reads per step
  constexpr int A_ST = BM * BK, B_ST = BN * BK, B_LDS = 2 * A_ST;              // halves
  constexpr bool MIDBAR = VAR & 32;
  constexpr int QB = MIDBAR ? NM / 2 : NM - 1;                                 // the barrier follows MFMA QB of step 2
  constexpr int RPS = (VAR & 64) ? 2 : 1;                                      // fragment reads per MFMA slot
  constexpr int WSEL = VAR & 3;
  constexpr int W3 = WSEL == 3 ? (QB + 1) / 2 : 0;                             // DMA slots used in step 2
  constexpr int WLEN = WSEL == 1 ? 2 * NM : WSEL == 2 ? NM : 3 * NM + W3;      // DMA window of a tile, in MFMA slots
  static_assert((BM / 8) % NW == 0 && (BN / 8) % NW == 0, "DMA chunks must divide over the waves");
  __shared__ __attribute__((aligned(16))) half_t smem[2 * A_ST + 2 * B_ST];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave % WGM, wn = wave / WGM;
  const int l32 = lane & 31, h = lane >> 5;

  const __amdgpu_buffer_rsrc_t rA =
      __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A - a_shift), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, b_bytes, 0x00020000);

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
    b_voff[j] = (unsigned)(n0 + r) * (unsigned)p.ldb * 2u + pk;      // N % 320 == 0: every row exists
  }

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
  // DMA instruction k of a wave's ND for K tile kt into stage stg: k < ACH an A chunk, else a B chunk
  auto dma = [&](auto kc, int kt, auto stgc) {
    constexpr int k = decltype(kc)::value, stg = decltype(stgc)::value;
    unsigned soa, sob;
    int tap;
    const bool live = kt < KT;
    ktile(live ? kt : 0, soa, sob, tap);
    if constexpr (k < ACH) {
      unsigned v = a_voff[k];
      if (MODE != MODE_DIRECT) v = ((a_mask[k] >> tap) & 1u) ? v : OOB;
      dma16(rA, &smem[stg * A_ST + (k * NW + wave) * 8 * BK], live ? v : OOB, soa);
    } else {
      constexpr int j = k - ACH;
      dma16(rB, &smem[B_LDS + stg * B_ST + (j * NW + wave) * 8 * BK], live ? b_voff[j] : OOB, sob);
    }
  };
  // the DMA instructions of window position g (0 .. WLEN - 1): instruction k sits at slot (k WLEN + WLEN / 2) / ND
  auto dma_slot = [&](auto gc, int kt, auto stgc) {
    constexpr int g = decltype(gc)::value;
    if constexpr (!(VAR & 8)) {
      static_for<ND>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr ((k * WLEN + WLEN / 2) / ND == g) dma(kc, kt, stgc);
      });
    }
  };

  float16_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  // fragment read offsets (halves) of step 0; step s flips piece bits 1-2: offset ^ (s << 4)
  const int key = (l32 >> 1) & 7;
  const int a_rd = (wm * MT * 32 + l32) * BK + ((h ^ key) << 3);
  const int b_rd = B_LDS + (wn * 160 + l32) * BK + ((h ^ key) << 3);

  half8_t xf[2][MT], wf[2][NT];
  // fragment read r (0 .. NR - 1) of (stage stg, step s) into register set `set`
  auto frag_read = [&](auto rc, auto setc, auto stgc, auto sc) {
    constexpr int r = decltype(rc)::value, set = decltype(setc)::value, stg = decltype(stgc)::value, s = decltype(sc)::value;
    constexpr int i = r == 0 ? 0 : r - NT;      // r = 0: xf[0] (every MFMA of the first row needs it); 1 .. NT: wf; then xf[1 ..]
    if constexpr (r >= 1 && r <= NT) wf[set][r - 1] = ld_half8(smem + ((b_rd ^ (s << 4)) + stg * B_ST + (r - 1) * 32 * BK));
    else xf[set][i] = ld_half8(smem + ((a_rd ^ (s << 4)) + stg * A_ST + i * 32 * BK));
  };

  // one 16-deep step: NM MFMAs out of set CUR; the reads of the next step (stage RSTG, step RS) into set CUR ^ 1; the DMA
  // instructions of window step W (0 .. 3, -1 = none) of K tile kt_dma into stage DSTG; SYNC: the tile's barrier
  auto step = [&](auto curc, auto rstgc, auto rsc, auto wc, int kt_dma, auto dstgc, auto syncc) {
    constexpr int cur = decltype(curc)::value, W = decltype(wc)::value;
    constexpr bool sync = decltype(syncc)::value != 0;
    constexpr int rps = (sync && MIDBAR) ? 2 : RPS;
    static_for<NM>([&](auto qc) {
      constexpr int q = decltype(qc)::value;
      constexpr int i = q / NT, j = q % NT;
      acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[cur][j], xf[cur][i], acc[i][j], 0, 0, 0);
      if constexpr (!(VAR & 16)) {
        static_for<rps>([&](auto uc) {
          constexpr int r = q * rps + decltype(uc)::value;
          if constexpr (r < NR) frag_read(ic<r>{}, ic<cur ^ 1>{}, rstgc, rsc);
        });
      }
      if constexpr (W >= 0 && W * NM + q < WLEN) dma_slot(ic<W * NM + q>{}, kt_dma, dstgc);
      G9_PIN();
      if constexpr (sync && q == QB) {
        // every fragment read of this tile has returned; this wave's share of tile t + 1 has landed
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        bar();
        if constexpr (VAR & 4) {
          if (wave >= NW / 2) __builtin_amdgcn_s_sleep(2);
        }
        G9_PIN();
      }
    });
  };

  // one K tile in stage STG: steps 0-2 carry window steps 1-3 of tile t + 1 (other stage), step 3 window step 0 of tile
  // t + 2 (this stage, free after the barrier) and the first fragment reads of tile t + 1
  auto tile = [&](auto stgc, int t) {
    constexpr int stg = decltype(stgc)::value;
    step(ic<0>{}, stgc, ic<1>{}, ic<1>{}, t + 1, ic<stg ^ 1>{}, ic<0>{});
    step(ic<1>{}, stgc, ic<2>{}, ic<2>{}, t + 1, ic<stg ^ 1>{}, ic<0>{});
    step(ic<0>{}, stgc, ic<3>{}, ic<3>{}, t + 1, ic<stg ^ 1>{}, ic<1>{});
    step(ic<1>{}, ic<stg ^ 1>{}, ic<0>{}, ic<0>{}, t + 2, stgc, ic<0>{});
  };

  // prologue: tile 0 landed and published, its step-0 fragments in set 0, window step 0 of tile 1 issued
  static_for<ND>([&](auto kc) { dma(kc, 0, ic<0>{}); });
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  bar();
  G9_PIN();
  static_for<NR>([&](auto rc) { frag_read(rc, ic<0>{}, ic<0>{}, ic<0>{}); });
  static_for<(NM < WLEN ? NM : WLEN)>([&](auto qc) { dma_slot(qc, 1, ic<1>{}); });
  G9_PIN();
  for (int t = 0; t < KT; t += 2) {
    tile(ic<0>{}, t);
    if (t + 1 < KT) tile(ic<1>{}, t + 1);
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // zero-fill DMA of the tiles past the end

  // ---- epilogue --------------------------------------------------------------------------------------------------------
  // MFMA layout (weights as the A operand): lane (l32, h) holds for row m = m0 + 32 (wm MT + i) + l32 of fragment (i, j) the
  // columns 8 (r / 4) + 4 h + r % 4, r = 0 .. 15.  v_permlane32_swap between the two half-waves turns the four 4-column
  // pieces into two 8-column runs per lane: columns 16 pr + 8 h .. + 7 (pr = 0, 1) - 16-byte loads and stores.
  const bool relu = p.flags & SKG_EPI_RELU;
  const int mrow = m0 + wm * (MT * 32) + l32;
  const int ncol = n0 + wn * 160 + h * 8;
  half8_t bv[NT][2];
#pragma unroll
  for (int j = 0; j < NT; ++j)
#pragma unroll
    for (int pr = 0; pr < 2; ++pr) bv[j][pr] = p.bias ? ld_half8(p.bias + ncol + j * 32 + pr * 16) : zero_half8();
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = mrow + i * 32;
    const int mc = min(m, p.M - 1);
    half8_t rv[NT][2];
    if (p.res) {
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int pr = 0; pr < 2; ++pr) rv[j][pr] = ld_half8(p.res + (size_t)mc * p.ldr + ncol + j * 32 + pr * 16);
    }
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      float16_t v = acc[i][j];
#pragma unroll
      for (int pr = 0; pr < 2; ++pr)
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          // (scalar copies on both sides: __builtin_bit_cast applied to an ext-vector ELEMENT reads element 0 of the vector -
          // hipcc / ROCm 7.2; seen as outputs 8x + 4 .. 7 repeating outputs 8x .. 8x + 3)
          const float lo4 = v[8 * pr + e], hi4 = v[8 * pr + 4 + e];
          const auto sw = __builtin_amdgcn_permlane32_swap(__builtin_bit_cast(unsigned, lo4), __builtin_bit_cast(unsigned, hi4),
                                                           false, false);
          const unsigned s0 = sw[0], s1 = sw[1];
          v[8 * pr + e] = __builtin_bit_cast(float, s0);
          v[8 * pr + 4 + e] = __builtin_bit_cast(float, s1);
        }
#pragma unroll
      for (int pr = 0; pr < 2; ++pr) {
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          float x = (v[8 * pr + e] + (float)bv[j][pr][e]) * p.alpha;
          if (p.res) x += (float)rv[j][pr][e];
          if (relu) x = fmaxf(x, 0.f);
          o[e] = (half_t)x;
        }
        if (m < p.M) st_half8(reinterpret_cast<half_t*>(p.C) + (size_t)m * p.ldc + ncol + j * 32 + pr * 16, o);
      }
    }
  }
}

inline bool operand_bytes9(const GemmParams& p, int mode, unsigned long long& a, unsigned long long& b,
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

// SKG_GEMM9 (read once): 0 / unset = off; 100 c + v = geometry c (1: <4, 2>, 3: <2, 2>) with schedule variant v
int gemm9_mode() {
  static const int v = getenv("SKG_GEMM9") ? atoi(getenv("SKG_GEMM9")) : 0;
  return v;
}

// rows of the block tile v9 would use for this launch (0 = not taken)
int gemm9_tile(const GemmParams& p, int mode) {
  const int md = gemm9_mode();
  if (!md) return 0;
  if (mode != MODE_DIRECT && mode != MODE_S1) return 0;
  if (p.K % BK != 0 || p.K < 2 * BK || p.M < 1 || p.N % BN != 0) return 0;
  if (mode == MODE_S1 && p.Cin % BK != 0) return 0;
  if (p.flags & (SKG_EPI_OUT_F32 | SKG_EPI_GEGLU)) return 0;
  if (p.c_lo || p.res_lo) return 0;
  if (p.ntaps || p.up2 || p.seg_rows || p.K2) return 0;
  if (p.ldc % 8 != 0 || (p.res && p.ldr % 8 != 0)) return 0;
  if (!skg_aligned(p.C, 16) || (p.res && !skg_aligned(p.res, 16)) || (p.bias && !skg_aligned(p.bias, 16))) return 0;
  unsigned long long a, b, s;
  if (!operand_bytes9(p, mode, a, b, s)) return 0;
  return md / 100 == 3 ? 128 : 256;
}

template <int WGM, int MT, int MODE>
void launch9v(const GemmParams& p, int var, int ntiles, int tiles_n, unsigned a, unsigned b, unsigned s, hipStream_t st) {
#define G9_CASE(V_)                                                                                                     \
  case V_:                                                                                                              \
    hipLaunchKernelGGL((gemm9_kernel<WGM, MT, MODE, V_>), dim3(ntiles), dim3(WGM * 128), 0, st, p, tiles_n, ntiles, a, b, s); \
    break
  switch (var) {
    G9_CASE(1);
    G9_CASE(2);
#ifdef SKG_LAB
    G9_CASE(4);
    G9_CASE(5);
    G9_CASE(6);
    G9_CASE(64);
    G9_CASE(65);
    G9_CASE(8);
    G9_CASE(16);
    G9_CASE(24);
#endif
    default:
      hipLaunchKernelGGL((gemm9_kernel<WGM, MT, MODE, 0>), dim3(ntiles), dim3(WGM * 128), 0, st, p, tiles_n, ntiles, a, b, s);
      break;
  }
#undef G9_CASE
}

template <int MODE>
void launch9(const GemmParams& p, hipStream_t st) {
  unsigned long long a, b, s;
  operand_bytes9(p, MODE, a, b, s);
  const int md = gemm9_mode(), geo = md / 100, var = md % 100;
  const int bm = geo == 3 ? 128 : 256;
  const int tiles_n = p.N / BN;
  const int ntiles = skg_cdiv(p.M, bm) * tiles_n;
  if (geo == 3) launch9v<2, 2, MODE>(p, var, ntiles, tiles_n, (unsigned)a, (unsigned)b, (unsigned)s, st);
  else launch9v<4, 2, MODE>(p, var, ntiles, tiles_n, (unsigned)a, (unsigned)b, (unsigned)s, st);
}

}  // namespace

bool skg_gemm9_eligible(const GemmParams& p, int mode) { return gemm9_tile(p, mode) != 0; }

bool skg_gemm9_try_launch(const GemmParams& p, int mode, hipStream_t st) {
  if (!gemm9_tile(p, mode)) return false;
  if (mode == MODE_DIRECT) launch9<MODE_DIRECT>(p, st);
  else launch9<MODE_S1>(p, st);
  return true;
}

// WITHDRAWN EXPERIMENT (round 3; lab build only - `make -C sketch2img_amd/csrc lab`, SKG_GEMMK=1).  Measured on MI355X
// with tools/smallm_bench.py (profiles/r03_smallm_kpair_probes.txt): 5-20 % SLOWER than gemm2.hip on the launches it was
// built for (bucket-weighted GEMM 116 vs 106 ms, convolution 189 vs 165 ms per batch; end to end 5.98 vs 6.20 images/s),
// although its own no-DMA probe runs the same launches at 1.2-1.9 PFLOP/s: the launches are bound by the bytes the
// L2 -> LDS path has to deliver per flop (128 x 160 tiles: 14 B per kFLOP whoever computes them), not by the latency a
// second wave per SIMD hides.  EXPERIMENTS.md, round 3.
//
// "k-pair" fp16 MFMA GEMM / 3x3 implicit-GEMM convolution for gfx950: the launches that put AT MOST ONE 128 x 160 tile
// on a CU (the 16x16 / 8x8-resolution layers and the cond-only backward: M = 512 ... 8192).
//
//   C[m][n] = epi(alpha * (sum_k A[m][k] * B[n][k] + bias[n]) + residual[m][n])
//
// Why: with <= 256 tiles gemm2.hip has ONE 4-wave workgroup per CU - one wave per SIMD, nobody to
// cover its LDS-DMA waits, fragment-read latency and barriers: the K loop ran at ~2 000 cycles per 64-deep step against
// 640 cycles of MFMA issue (M4096 N1280 K1280: 25.7 us for 5.4 us of matrix time), or, split over K, paid fp32 slabs and
// a second launch.  Two co-resident workgroups is what makes the same kernel reach ~1.1 PFLOP/s on big layers, and a
// small layer has no second tile to give - but it has a second HALF OF K.
//
// Here ONE 8-wave workgroup owns the 128 x 160 tile and its two wave groups (waves 0-3 / 4-7; waves w and w + 4 share a
// SIMD) take ALTERNATE 64-deep K tiles: group g accumulates K tiles 2t + g over the whole tile (2 x 2 waves of 64 x 80,
// the wave tile of gemm2.hip: 9 fragment reads per 20 MFMAs).  The two groups run the ping-pong schedule of gemm8.hip -
//      LOAD(j): 9 fragment reads of k-sub-step j (+ the 9 LDS-DMA instructions of the next K-tile pair) -> s_barrier ->
//      COMPUTE(j): 20 MFMAs out of registers at raised priority                                         -> s_barrier
// one barrier apart, so on every SIMD one wave streams MFMAs while its partner reads LDS and issues DMA.  At the end the
// partial sums of group 1 cross to group 0 through LDS (the dead pipeline stages: no HBM round trip, no second launch)
// and the epilogue runs once.  In LDS a "stage" holds BOTH groups' K tiles as one virtual (256 + 320)-row x 64 tile
// (rows 128.. of A / 160.. of B = the same rows, next K tile), i.e. exactly the image of gemm8.hip's 256 x 320 tile:
// two stages of 73.7 KB, the whole next pair requested in LOAD(2t) and waited for with vmcnt(0) at the end of LOAD(2t + 1).
//
// Launches with too few tiles even for that (8x8 level: 32-128 tiles, K up to 23 040) still split K ACROSS workgroups
// (fp32 slabs + splitk_reduce), but over half as many slices as before: the in-workgroup pair is the first factor 2.
// Scope: MODE_DIRECT and MODE_S1, K % 64 == 0 (conv: Cin % 64 == 0), N % 160 == 0, fp16 output, no fused GEGLU, no
// GroupNorm statistics; everything else stays on gemm2.hip.
#include "gemm_params.h"
#include <stdlib.h>

namespace {

constexpr int BM = 128, BN = 160, BK = 64, NW = 8, NTHR = 512;
constexpr int MT = 4, NT = 5;                    // wave tile 64 x 80
constexpr int ACH = 4, BCH = 5;                  // 8-row chunks per wave: 32 of A', 40 of B' over 8 waves (9 DMA instructions per step)
constexpr int VA = 2 * BM, VB = 2 * BN;          // virtual rows of a stage: both groups' K tiles
constexpr int STAGE = (VA + VB) * BK;            // halves per stage (73 728 B)
constexpr int OPF = BN + 4;                      // pitch (floats) of the fp32 exchange / staging slab
constexpr unsigned OOB = 0x80000000u;
static_assert(BM * OPF * 4 <= 2 * STAGE * 2, "the exchange slab lives in the dead stages");

typedef __attribute__((address_space(3))) void* lds_ptr_t;

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, half_t* lds_wave_base, unsigned voff, unsigned soff) {
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ void bar() { asm volatile("s_barrier" ::: "memory"); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// EXP (lab build only, SKG_GK_EXP; compute on stale / missing data, timing only): 1 = no DMA in the loop, 2 = no global
// stores in the epilogue, 4 = no stagger between the two groups
template <int MODE, int EXP = 0>
__global__ __launch_bounds__(NTHR, 2) void gemmk_kernel(const GemmParams p, int tiles_n, int nwg, unsigned a_bytes,
                                                        unsigned b_bytes, unsigned a_shift, int kt_per_split,
                                                        float* __restrict__ ws) {
  __shared__ __attribute__((aligned(16))) half_t smem[2 * STAGE + 2 * BN];     // two stages + the tile's fp32 bias slice
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2;                       // waves w and w + 4 share a SIMD: one of each group
  const int wm = (wave >> 1) & 1, wn = wave & 1;   // 2 x 2 waves inside a group
  const int gq = lane >> 4, l16 = lane & 15;

  const __amdgpu_buffer_rsrc_t rA =
      __builtin_amdgcn_make_buffer_rsrc((void*)((const char*)p.A - a_shift), 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc((void*)p.B, 0, b_bytes, 0x00020000);

  // ---- XCD-aware tile assignment (workgroup b -> XCD b % 8), as gemm2.hip ---------------------------------------------
  int lid;
  {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, idx = bid >> 3;
    const int q = nwg >> 3, r = nwg & 7;
    lid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  }
  const int KTall = p.K / BK;
  const int ntiles = ws ? nwg / ((KTall + kt_per_split - 1) / kt_per_split) : nwg;
  const int split = lid / ntiles;
  lid -= split * ntiles;
  int tile_m = lid / tiles_n;
  int tile_n = lid - tile_m * tiles_n;
  if (const int gn = (p.flags >> 20) & 0xf) {      // host-chosen gm x gn arrangement of the XCDs over the tile grid
    const int G = ((p.flags >> 24) & 0xf) ? (int)((p.flags >> 24) & 0xf) : 8;
    const int per = ntiles / G, tmb = (ntiles / tiles_n) / (G / gn), tnb = tiles_n / gn;
    const int xr = lid / per, r = lid - xr * per;
    const int xm = xr / gn, xn = xr - xm * gn;
    const int rm = r / tnb;
    tile_m = xm * tmb + rm;
    tile_n = xn * tnb + (r - rm * tnb);
  }
  const int m0 = tile_m * BM, n0 = tile_n * BN;
  const int kt_begin = split * kt_per_split;
  const int kt_end = min(KTall, kt_begin + kt_per_split);
  const int nst = (kt_end - kt_begin + 1) >> 1;    // super-steps: pairs of K tiles

  // ---- per-lane DMA source description: one voffset per operand row, K-tile / tap position in the soffset --------------
  const int lr = lane >> 3, lq = lane & 7;
  unsigned a_voff[ACH], a_mask[ACH];
#pragma unroll
  for (int j = 0; j < ACH; ++j) {
    const int vr = (j * NW + wave) * 8 + lr;       // virtual row: j < 2 -> group 0's K tile, j >= 2 -> group 1's
    const int r = vr & (BM - 1);
    const unsigned pk = (unsigned)(lq ^ ((vr >> 1) & 7)) * 16u;
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
  // B' chunk c = j * 8 + wave covers virtual rows 8c .. 8c + 7; rows >= 160 (c >= 20) are group 1's K tile
  const int b2_half = __builtin_amdgcn_readfirstlane((2 * NW + wave) >= BN / 8 ? 1 : 0);      // j = 2: waves 4-7
#pragma unroll
  for (int j = 0; j < BCH; ++j) {
    const int vr = (j * NW + wave) * 8 + lr;
    const int r = vr >= BN ? vr - BN : vr;
    const unsigned pk = (unsigned)(lq ^ ((vr >> 1) & 7)) * 16u;
    b_voff[j] = (unsigned)(n0 + r) * (unsigned)p.ldb * 2u + pk;
  }

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
  // the ND LDS-DMA instructions of super-step t (K tiles kt_begin + 2t and + 2t + 1) into stage `buf`; a K tile beyond
  // the range (odd tile count, or the step behind the last) is an out-of-range zero fill
  auto dma_issue = [&](int t, int buf) {
    unsigned soa[2], sob[2];
    int tap[2];
    bool live[2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int kt = kt_begin + 2 * t + h;
      live[h] = kt < kt_end;
      ktile(live[h] ? kt : kt_begin, soa[h], sob[h], tap[h]);
    }
#pragma unroll
    for (int j = 0; j < ACH; ++j) {
      const int h = j >> 1;
      unsigned v = a_voff[j];
      if (MODE != MODE_DIRECT) v = ((a_mask[j] >> tap[h]) & 1u) ? v : OOB;
      dma16(rA, &smem[buf * STAGE + (j * NW + wave) * 8 * BK], live[h] ? v : OOB, soa[h]);
    }
#pragma unroll
    for (int j = 0; j < BCH; ++j) {
      const bool hi = j > 2 || (j == 2 && b2_half);          // wave-uniform
      const bool lv = hi ? live[1] : live[0];
      dma16(rB, &smem[buf * STAGE + VA * BK + (j * NW + wave) * 8 * BK], lv ? b_voff[j] : OOB, hi ? sob[1] : sob[0]);
    }
  };

  float4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};

  // fragment read addresses (halves) inside a stage for k-sub-step 0 / 1; slot = piece ^ ((row >> 1) & 7).  Every wave
  // tile starts at a multiple of 16 virtual rows, so the swizzle key is (l16 >> 1) & 7 for every fragment: ONE address
  // per operand and sub-step, the fragment index is an immediate offset.
  const int key = (l16 >> 1) & 7;
  int a_base[2], b_base[2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks) {
    a_base[ks] = (grp * BM + wm * 64 + l16) * BK + (((ks * 4 + gq) ^ key) << 3);
    b_base[ks] = VA * BK + (grp * BN + wn * 80 + l16) * BK + (((ks * 4 + gq) ^ key) << 3);
  }

  half8_t xf[MT], wf[NT];
  // one sub-step of one wave: LOAD segment, barrier, COMPUTE segment, barrier
  auto substep = [&](int t, int stg, int ks) {        // stg, ks: compile-time constants at every call site
    const half_t* sb = &smem[stg * STAGE];
#pragma unroll
    for (int j = 0; j < NT; ++j) wf[j] = ld_half8(sb + b_base[ks] + j * 16 * BK);
#pragma unroll
    for (int i = 0; i < MT; ++i) xf[i] = ld_half8(sb + a_base[ks] + i * 16 * BK);
    if (!(EXP & 1) && ks == 0) dma_issue(t + 1, stg ^ 1);      // the other stage's last reader: group 1's LOAD(2t - 1), one barrier ago
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (!(EXP & 1) && ks == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the pair LOAD(2t + 2) reads has landed
    __builtin_amdgcn_sched_barrier(0);
    bar();
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_setprio(1);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
    __builtin_amdgcn_sched_barrier(0);
    bar();
    __builtin_amdgcn_sched_barrier(0);
  };

  // the tile's bias slice and the residual pieces this thread finishes in the epilogue are requested now (phase-2 map:
  // thread -> row er = tid / 4, pieces (tid % 4) + 4k of its 20): their latency hides under the K loop
  // (N % 160 == 0: every column of the tile exists.)  The bias too comes through a descriptor and stays a raw half until
  // the epilogue: a conditional load whose value is converted at once parks the wave for a round trip before the first DMA
  const bool has_bias = !ws && p.bias;
  const __amdgpu_buffer_rsrc_t rBias = __builtin_amdgcn_make_buffer_rsrc((void*)(has_bias ? p.bias : p.A), 0,
                                                                          has_bias ? (unsigned)p.N * 2u : 0u, 0x00020000);
  const unsigned short bias_raw = __builtin_amdgcn_raw_buffer_load_b16(rBias, tid < BN ? (unsigned)(n0 + tid) * 2u : OOB, 0, 0);
  const int er = tid >> 2, ec = (tid & 3) * 8;
  const bool rowok = m0 + er < p.M;
  // through a buffer descriptor: rows beyond M (and every piece when there is no residual: a zero-sized range) read as
  // zeros by the range check, so the five loads are unconditional and back to back (a per-piece "load or zero" select makes
  // hipcc branch around every load and wait vmcnt(0) for each: five serial HBM round trips before the first K tile)
  const bool has_res = !ws && p.res;
  const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc(
      (void*)(has_res ? p.res : p.A), 0, has_res ? (unsigned)(((size_t)(p.M - 1) * p.ldr + p.N) * 2) : 0u, 0x00020000);
  half8_t rv[NT];
  {
    const unsigned ro = rowok ? (unsigned)(((size_t)(m0 + er) * p.ldr + n0 + ec) * 2) : OOB;
#pragma unroll
    for (int k = 0; k < NT; ++k) rv[k] = __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rR, ro, k * 64, 0));
  }

  // prologue: the first pair in flight, landed and published
  dma_issue(0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  bar();
  if (grp == 1 && !(EXP & 4)) bar();                   // group 1 runs one barrier behind group 0
  for (int t = 0; t < nst; t += 2) {
    substep(t, 0, 0);
    substep(t, 0, 1);
    if (t + 1 < nst) {
      substep(t + 1, 1, 0);
      substep(t + 1, 1, 1);
    }
  }
  if (grp == 0 && !(EXP & 4)) bar();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the last step's zero-fill DMA must not land in the exchange slab

  // ---- epilogue ------------------------------------------------------------------------------------------------------
  // lane holds C[m = wm*64 + 16 i + l16][n = wn*80 + 16 j + 4 gq .. + 3] of ITS GROUP's half of K.  Group 1 parks its
  // partial sums in LDS, group 0 adds its own (and the bias) IN PLACE - every lane touches only its own positions - and
  // the slab is then the fp32 staging slab of gemm2.hip's epilogue: all 512 threads walk whole output rows with 16-byte
  // residual loads / 16-byte stores, alpha, residual and ReLU in fp32, one rounding.
  float* const X = reinterpret_cast<float*>(smem);
  float* const bias_s = reinterpret_cast<float*>(smem + 2 * STAGE);
  if (tid < BN) bias_s[tid] = (float)__builtin_bit_cast(half_t, bias_raw);
  lds_barrier();                                       // every wave is past its last fragment read
  float* const xw = X + (wm * 64 + l16) * OPF + wn * 80 + gq * 4;
  if (grp == 1) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j) *reinterpret_cast<float4_t*>(xw + i * 16 * OPF + j * 16) = acc[i][j];
  }
  lds_barrier();
  if (grp == 0) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const float4_t b4 = *reinterpret_cast<const float4_t*>(&bias_s[wn * 80 + j * 16 + gq * 4]);
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        float4_t* q = reinterpret_cast<float4_t*>(xw + i * 16 * OPF + j * 16);
        *q = *q + acc[i][j] + b4;
      }
    }
  }
  lds_barrier();
  if (!rowok) return;
  const float* const srow = X + er * OPF + ec;
  float4_t v0[NT], v1[NT];
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    v0[k] = *reinterpret_cast<const float4_t*>(srow + k * 32);
    v1[k] = *reinterpret_cast<const float4_t*>(srow + k * 32 + 4);
  }
  if (ws) {      // split-K partial across workgroups: raw fp32 sums, the epilogue happens in splitk_reduce
    float* const slab = ws + (size_t)split * p.M * p.N + (size_t)(m0 + er) * p.N + n0 + ec;
#pragma unroll
    for (int k = 0; k < NT; ++k) {
      *reinterpret_cast<float4_t*>(slab + k * 32) = v0[k];
      *reinterpret_cast<float4_t*>(slab + k * 32 + 4) = v1[k];
    }
    return;
  }
  const bool relu = p.flags & SKG_EPI_RELU;
  const bool stream_out = p.flags & 0x800u;
  half_t* const crow = reinterpret_cast<half_t*>(p.C) + (size_t)(m0 + er) * p.ldc + n0 + ec;
#pragma unroll
  for (int k = 0; k < NT; ++k) {
    float v[8] = {v0[k][0], v0[k][1], v0[k][2], v0[k][3], v1[k][0], v1[k][1], v1[k][2], v1[k][3]};
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = v[e] * p.alpha + (float)rv[k][e];
    if (relu) {
      asm volatile("" ::: "memory");       // a real (wave-uniform) branch instead of 8 selects per piece
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e], 0.f);
    }
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)v[e];
    half8_t* dst8 = reinterpret_cast<half8_t*>(crow + k * 32);
    if (EXP & 2) { if (o[0] == (half_t)12345.f) *dst8 = o; continue; }       // probe: epilogue without stores
    // (the s_nop: two wait states between a > 8-byte VMEM store and a VALU write of its data registers - gemm2.hip)
    if (stream_out) asm volatile("global_store_dwordx4 %0, %1, off nt\n\ts_nop 1" ::"v"(dst8), "v"(o) : "memory");
    else *dst8 = o;
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

// Tuning knobs (environment, read once; defaults = what ships):
//   SKG_GEMMK       0 = off (every launch back on gemm2.hip: same-box A/B)
//   SKG_GK_SPLIT_KT minimum number of 64-deep K tiles per cross-workgroup slice (default 32: 16 super-steps)
int knob(const char* name, int dflt) {
  const char* v = getenv(name);
  return v ? atoi(v) : dflt;
}
// (read at every launch, not cached: tools/smallm_bench.py flips them inside one process for interleaved A/B rounds)
bool gk_on() { return knob("SKG_GEMMK", 0) != 0; }       // lab build: off unless asked for
int gk_split_kt() { const int v = knob("SKG_GK_SPLIT_KT", 32); return v < 2 ? 2 : v; }

// cross-workgroup K slices for `tiles` tiles of KT K-tiles: fill the 256 CUs, every slice >= gk_split_kt() tiles deep
int gk_splits(long tiles, int KT, size_t slab_bytes, const float* ws, size_t ws_bytes) {
  if (!ws || tiles > 128) return 1;
  int s = (int)(256 / tiles);
  if (s > 8) s = 8;
  while (s > 1 && KT / s < gk_split_kt()) --s;
  while (s > 1 && (size_t)s * slab_bytes > ws_bytes) --s;
  return s;
}

bool gk_takes(const GemmParams& p, int mode) {
  if (!gk_on()) return false;
  if (mode != MODE_DIRECT && mode != MODE_S1) return false;
  if (p.K % BK != 0 || p.K < 4 * BK || p.M < 1 || p.N % BN != 0) return false;
  if (mode == MODE_S1 && p.Cin % BK != 0) return false;
  if (p.flags & (SKG_EPI_OUT_F32 | SKG_EPI_GEGLU)) return false;
  if (p.gn_partial || p.aux) return false;
  if (p.ldc % 8 != 0 || (reinterpret_cast<uintptr_t>(p.C) & 15) != 0) return false;
  if (p.res && (p.ldr % 8 != 0 || (reinterpret_cast<uintptr_t>(p.res) & 15) != 0)) return false;
  unsigned long long a, b, s;
  if (!operand_bytes(p, mode, a, b, s)) return false;
  const long tiles = (long)skg_cdiv(p.M, BM) * (p.N / BN);
  return tiles <= 256;      // more: two workgroups per CU on gemm2.hip
}

}  // namespace

// splitk_reduce_kernel lives in gemm2.hip
void skg_splitk_reduce_launch(const GemmParams& p, const float* ws, int splits, hipStream_t st);

bool skg_gemmk_eligible(const GemmParams& p, int mode) { return gk_takes(p, mode); }

bool skg_gemmk_try_launch(const GemmParams& p_in, int mode, hipStream_t st) {
  if (!gk_takes(p_in, mode)) return false;
  GemmParams p = p_in;
  unsigned long long a, b, s;
  operand_bytes(p, mode, a, b, s);
  const int tiles_n = p.N / BN, tiles_m = skg_cdiv(p.M, BM);
  const int ntiles = tiles_m * tiles_n;
  const int KT = p.K / BK;
  const size_t out_bytes = (size_t)p.M * p.N * 2;
  if (out_bytes >= ((size_t)32 << 20)) p.flags |= 0x800u;
  const int splits = gk_splits(ntiles, KT, (size_t)p.M * p.N * 4, p.ws, p.ws_bytes);
  const int per = skg_cdiv(KT, splits);
  const int ns = skg_cdiv(KT, per);              // every slice non-empty
  // XCD arrangement for weight-heavy launches (see gemm2.hip): XCDs per K slice G = 8 / ns when that divides
  const int G = (8 % ns == 0) ? 8 / ns : 0;
  if (G >= 2 && ntiles % G == 0) {
    const double a_mb = (double)p.M * (mode == MODE_DIRECT ? p.K : p.Cin) * 2.0, w_mb = (double)p.N * p.K * 2.0;
    int best = 1;
    double cost = a_mb + G * w_mb;
    for (int gn = 2; gn <= G; gn *= 2) {
      if (tiles_n % gn != 0 || tiles_m % (G / gn) != 0) continue;
      const double c = gn * a_mb + (G / gn) * w_mb;
      if (c < 0.9 * cost) { cost = c; best = gn; }
    }
    if (best > 1) p.flags |= ((unsigned)best << 20) | ((unsigned)(G == 8 ? 0 : G) << 24);
  }
  float* slab = ns > 1 ? p.ws : nullptr;
#ifdef SKG_LAB
  const int exp = knob("SKG_GK_EXP", 0);
#else
  constexpr int exp = 0;
#endif
#define GK_LAUNCH(M_, E_) hipLaunchKernelGGL((gemmk_kernel<M_, E_>), dim3(ntiles * ns), dim3(NTHR), 0, st, p, tiles_n, \
                                             ntiles * ns, (unsigned)a, (unsigned)b, (unsigned)s, ns > 1 ? per : KT, slab)
  if (mode == MODE_DIRECT) {
    switch (exp) {
#ifdef SKG_LAB
      case 1: GK_LAUNCH(MODE_DIRECT, 1); break;
      case 2: GK_LAUNCH(MODE_DIRECT, 2); break;
      case 3: GK_LAUNCH(MODE_DIRECT, 3); break;
      case 4: GK_LAUNCH(MODE_DIRECT, 4); break;
#endif
      default: GK_LAUNCH(MODE_DIRECT, 0); break;
    }
  } else {
    switch (exp) {
#ifdef SKG_LAB
      case 1: GK_LAUNCH(MODE_S1, 1); break;
      case 2: GK_LAUNCH(MODE_S1, 2); break;
      case 3: GK_LAUNCH(MODE_S1, 3); break;
      case 4: GK_LAUNCH(MODE_S1, 4); break;
#endif
      default: GK_LAUNCH(MODE_S1, 0); break;
    }
  }
#undef GK_LAUNCH
  if (ns > 1) skg_splitk_reduce_launch(p, slab, ns, st);
  return true;
}

// WITHDRAWN EXPERIMENT (round 4; lab build only - `make -C sketch2img_amd/csrc lab`; tools/lab/rowgemm_bench.py measures and checks it).
// Correct (rel 2e-4 vs fp32, 99.9 % of the outputs bit-equal to skg_gemm_f16) and 7-30 % SLOWER than the gemm2.hip launches it was
// meant to replace: EXPERIMENTS.md round 4, "a short-K GEMM with resident weights".
//
// Short-K GEMM with the WEIGHTS resident in LDS and the activation rows streamed through registers:
//   Y[m][n] = sum_k LN?(X)[m][k] . W[n][k] + bias[n] (+ residual[m][n]),   K = 320 (the C = 320 transformer blocks of SD1.5's 64 x 64 level)
// replaces gemm2.hip's 128 x 160 tile on the launches that are nothing but an HBM stream with a 5-step K loop (attn1.to_out +
// residual, proj_in, the fused q / k / v projection, [diffusers] attention.py BasicTransformerBlock / Transformer2DModel, reached
// from modules/pipeline.py:96).  There the two-stage tile pipeline has ONE 36 KB operand stage in flight per workgroup - of which
// 20 KB are weights that come from L2 anyway - and a workgroup lives ~20 us for 1.5 us of matrix work: 3.0 TB/s on the + residual
// launch, 2.4-2.6 TB/s on the others (profiles/r04_cfg2_shapes.txt).
//
// Here a workgroup owns a 160-column slice of W for its whole life: 100 fragment-major pieces of 1 KB (the pack IS the LDS image,
// one conflict-free ds_read_b128 per MFMA), fetched once.  Every wave then walks 16-row blocks of X on its own - same transposed
// formulation as ffblock.hip / xattn.hip: the block's rows are the B operands, loaded straight from global memory into registers
// (10 x 16 bytes per lane), Y^T[160 x 16] = W_slice . X^T in 100 MFMAs (k-step outer, 10 independent accumulators that start from
// the bias), the residual added from lane-local 8-byte reads.  The NEXT block's rows and residual are already in flight while this
// one computes (register double buffer): 15 KB per wave, 120 KB per CU of pure activation traffic in flight, no barrier and no
// s_waitcnt vmcnt(0) anywhere in the steady state.  LayerNorm (optional) is applied to the rows in registers (norms.hip's two-pass
// form; slice 0 writes the statistics the backward reads), so norm1 + the q / k / v projection is one launch.
#include "common.h"

namespace {

typedef __attribute__((address_space(3))) void* lds_ptr_t;

struct RGParams {
  const half_t* X; int ldx;
  half_t* Y; int ldy;
  int M, N;
  const half_t* Wp;        // [N / 16][10][512]: piece (u, ks): [lane = 16 g + l][i] = W[16 u + l][32 ks + 8 g + i]
  const half_t* bias;      // [N] or nullptr
  const half_t* res; int ldr;        // optional residual [M][>= N]
  const half_t* gamma; const half_t* beta; float eps;      // optional LayerNorm of the rows of X (gamma != nullptr)
  float* stats;            // optional [M][2]: LayerNorm (mean, rstd)
  int nslices, wgs_per_slice;
};

template <bool LN, bool RES, int PROBE = 0>
__global__ __launch_bounds__(512, 1) void rowgemm_kernel(const RGParams p) {
  constexpr int KS = 10, C = 320, NT = 10, PIECE = 512;      // NT tiles of 16 output columns per slice
  __shared__ __attribute__((aligned(16))) half_t smem[NT * KS * PIECE];      // 100 KB: the slice of W, resident

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l16 = lane & 15, g = lane >> 4;
  // workgroup -> (slice, j): the slices of one j share an XCD (blockIdx & 7), so the rows they all read meet in one L2
  int slice, j;
  {
    const int id = blockIdx.x, xcd = id & 7, r = id >> 3;
    slice = r % p.nslices;
    j = (r / p.nslices) * 8 + xcd;
  }
  const int n0 = slice * (NT * 16);
  {   // the slice's pieces: 100 x 1 KB, 13 slots per wave (dead slots read out of range of a 0-byte descriptor... simply skipped)
    const __amdgpu_buffer_rsrc_t rW = __builtin_amdgcn_make_buffer_rsrc((void*)(p.Wp + (size_t)slice * NT * KS * PIECE), 0,
                                                                       NT * KS * PIECE * 2, 0x00020000);
#pragma unroll
    for (int s = 0; s < 13; ++s) {
      const int q = wave + 8 * s;
      if (q < NT * KS)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rW, (lds_ptr_t)(smem + q * PIECE), 16, (unsigned)lane * 16u, (unsigned)q * 1024u, 0, 0);
    }
  }
  half4_t bq[NT];          // the slice's bias, lane (l, g): columns 16 u + 4 g .. + 3
#pragma unroll
  for (int u = 0; u < NT; ++u) bq[u] = p.bias ? ld_half4(p.bias + n0 + 16 * u + 4 * g) : half4_t{0, 0, 0, 0};

  const int nb = (p.M + 15) >> 4;                       // 16-row blocks
  const int stride = p.wgs_per_slice * 8;
  int b = j * 8 + wave;

  half8_t xa[KS], xn[KS];
  half4_t ra[NT], rn[NT];
  auto fetch = [&](int blk, half8_t (&x)[KS], half4_t (&r)[NT]) {
    const int m = min(blk * 16 + l16, p.M - 1);
    const half_t* xr = p.X + (size_t)m * p.ldx + 8 * g;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) x[ks] = (PROBE == 3) ? half8_t{1, 2, 3, 4, 5, 6, 7, (half_t)(float)blk} : ld_half8(xr + 32 * ks);
    if constexpr (RES) {
      const half_t* rr = p.res + (size_t)m * p.ldr + n0 + 4 * g;
#pragma unroll
      for (int u = 0; u < NT; ++u) r[u] = ld_half4(rr + 16 * u);
    }
  };
  auto compute = [&](int blk, half8_t (&x)[KS], half4_t (&r)[NT]) {
    const int mrow = blk * 16 + l16;
    if constexpr (LN) {
      float s = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < 8; ++i) s += (float)x[ks][i];
      s += __shfl_xor(s, 16, 64);
      s += __shfl_xor(s, 32, 64);
      const float mean = s * (1.f / C);
      float s2 = 0.f;
#pragma unroll
      for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int i = 0; i < 8; ++i) { const float d = (float)x[ks][i] - mean; s2 += d * d; }
      s2 += __shfl_xor(s2, 16, 64);
      s2 += __shfl_xor(s2, 32, 64);
      const float rstd = rsqrtf(s2 * (1.f / C) + p.eps);
      if (p.stats && slice == 0 && g == 0 && mrow < p.M) { p.stats[(size_t)mrow * 2] = mean; p.stats[(size_t)mrow * 2 + 1] = rstd; }
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
        const half8_t gv = ld_half8(p.gamma + 32 * ks + 8 * g), bv = ld_half8(p.beta + 32 * ks + 8 * g);
#pragma unroll
        for (int i = 0; i < 8; ++i) x[ks][i] = (half_t)(((float)x[ks][i] - mean) * rstd * (float)gv[i] + (float)bv[i]);
      }
    }
    float4_t acc[NT];
#pragma unroll
    for (int u = 0; u < NT; ++u) acc[u] = float4_t{(float)bq[u][0], (float)bq[u][1], (float)bq[u][2], (float)bq[u][3]};
    const half_t* fr = smem + lane * 8;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      if constexpr (PROBE == 1) {      // no fragment reads, no MFMAs: the rows still have to arrive
#pragma unroll
        for (int u = 0; u < NT; ++u) acc[u][u & 3] += (float)x[ks][u & 7];
        continue;
      }
      half8_t wf[NT];
#pragma unroll
      for (int u = 0; u < NT; ++u) wf[u] = ld_half8(fr + (u * KS + ks) * PIECE);
#pragma unroll
      for (int u = 0; u < NT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[u], x[ks], acc[u], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);      // (without it hipcc hoists all 100 fragment reads: 400 registers, 1.7 KB of scratch)
    }
    if (mrow < p.M) {
      half_t* yr = p.Y + (size_t)mrow * p.ldy + n0 + 4 * g;
#pragma unroll
      for (int u = 0; u < NT; ++u) {
        float4_t v = acc[u];
        if constexpr (RES) v += float4_t{(float)r[u][0], (float)r[u][1], (float)r[u][2], (float)r[u][3]};
        if (PROBE == 2) { if (v[0] == 12345.f) st_half4(yr + 16 * u, half4_t{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]}); continue; }
        st_half4(yr + 16 * u, half4_t{(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]});
      }
    }
  };

  if (b < nb) fetch(b, xa, ra);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the weight slice (and the first block) landed
  __syncthreads();
  // two blocks per trip: the registers of one are in flight while the other computes
  while (b < nb) {
    const int b1 = b + stride;
    if (b1 < nb) fetch(b1, xn, rn);
    compute(b, xa, ra);
    if (b1 >= nb) break;
    const int b2 = b1 + stride;
    if (b2 < nb) fetch(b2, xa, ra);
    compute(b1, xn, rn);
    b = b2;
  }
}

}  // namespace

static int rowgemm_impl(const void* X, int ldx, const void* Wpack, void* Y, int ldy, int M, int N, int K, const void* bias,
                        const void* residual, int ldr, const void* gamma, const void* beta, float eps, float* stats, void* stream) {
  SKG_REQUIRE(X && Wpack && Y && M > 0 && K == 320 && N > 0 && N % 160 == 0);
  SKG_REQUIRE(ldx % 8 == 0 && ldx >= K && ldy % 4 == 0 && ldy >= N && (!residual || (ldr % 4 == 0 && ldr >= N)));
  SKG_REQUIRE(skg_aligned(X, 16) && skg_aligned(Wpack, 16) && skg_aligned(Y, 8) && skg_aligned(bias, 8) && skg_aligned(residual, 8));
  SKG_REQUIRE((gamma != nullptr) == (beta != nullptr) && skg_aligned(gamma, 16) && skg_aligned(beta, 16) && (!stats || gamma));
  SKG_REQUIRE(X != Y);      // (another slice's workgroup may still be reading the rows this one stores)
  RGParams p;
  p.X = (const half_t*)X; p.ldx = ldx; p.Y = (half_t*)Y; p.ldy = ldy; p.M = M; p.N = N;
  p.Wp = (const half_t*)Wpack; p.bias = (const half_t*)bias; p.res = (const half_t*)residual; p.ldr = ldr;
  p.gamma = (const half_t*)gamma; p.beta = (const half_t*)beta; p.eps = eps; p.stats = stats;
  p.nslices = N / 160;
  // one workgroup per CU (100 KB of LDS), a multiple of 8 per slice (XCD grouping), never more waves than 16-row blocks
  int wgs = (256 / p.nslices) / 8 * 8;
  const int need = (skg_cdiv(M, 16) + 7) / 8;
  if (wgs > (need + 7) / 8 * 8) wgs = (need + 7) / 8 * 8;
  if (wgs < 8) wgs = 8;
  p.wgs_per_slice = wgs;
  const dim3 grid((unsigned)(wgs * p.nslices));
  hipStream_t st = (hipStream_t)stream;
#ifdef SKG_LAB      // cost probes, wrong results (EXPERIMENTS.md round 4): 1 no matrix work, 2 no stores, 3 no row loads
  if (const char* pr = getenv("SKG_RG_PROBE")) {
    const int v = atoi(pr);
    if (!gamma && residual) {
      if (v == 1) hipLaunchKernelGGL((rowgemm_kernel<false, true, 1>), grid, dim3(512), 0, st, p);
      else if (v == 2) hipLaunchKernelGGL((rowgemm_kernel<false, true, 2>), grid, dim3(512), 0, st, p);
      else hipLaunchKernelGGL((rowgemm_kernel<false, true, 3>), grid, dim3(512), 0, st, p);
      SKG_CHECK_LAUNCH("skg_rowgemm_f16 (probe)");
      return SKG_OK;
    }
  }
#endif
  if (gamma) {
    if (residual) hipLaunchKernelGGL((rowgemm_kernel<true, true>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((rowgemm_kernel<true, false>), grid, dim3(512), 0, st, p);
  } else {
    if (residual) hipLaunchKernelGGL((rowgemm_kernel<false, true>), grid, dim3(512), 0, st, p);
    else hipLaunchKernelGGL((rowgemm_kernel<false, false>), grid, dim3(512), 0, st, p);
  }
  SKG_CHECK_LAUNCH("skg_rowgemm_f16");
  return SKG_OK;
}

extern "C" int skg_rowgemm_f16(const void* X, int ldx, const void* Wpack, void* Y, int ldy, int M, int N, int K, const void* bias,
                               const void* residual, int ldr, void* stream) {
  return rowgemm_impl(X, ldx, Wpack, Y, ldy, M, N, K, bias, residual, ldr, nullptr, nullptr, 0.f, nullptr, stream);
}

extern "C" int skg_ln_rowgemm_f16(const void* X, int ldx, const void* Wpack, void* Y, int ldy, int M, int N, int K, const void* bias,
                                  const void* residual, int ldr, const void* gamma, const void* beta, float eps, float* stats,
                                  void* stream) {
  SKG_REQUIRE(gamma && beta);
  return rowgemm_impl(X, ldx, Wpack, Y, ldy, M, N, K, bias, residual, ldr, gamma, beta, eps, stats, stream);
}

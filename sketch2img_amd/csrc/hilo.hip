// Accuracy mode ("residual_fp32"): GroupNorm-apply and LayerNorm whose INPUT is a (hi, lo) pair of fp16 tensors, x = hi + lo
// (~22 mantissa bits: what the hi / lo epilogue of gemm2.hip writes).  north_star asks for <= 1e-3 max eps deviation from
// the fp32 reference; an fp16 residual stream alone costs 1.0e-3 (tools/eps_decompose.py), so HipUNet's opt-in
// accuracy mode keeps the residual stream and the conv outputs that feed a norm as pairs.  Own, simple kernels: the tuned
// fp16 ones in norms.hip stay untouched.  Statistics come from the hi part alone through the existing entry point
// (skg_groupnorm_stats): the mean of n >= 640 rounding errors of relative size 2^-12 is far below fp32 resolution.
#include "common.h"

namespace {

// workgroup = (pixel chunk, row); thread = fixed 8-channel piece x strided pixels (as gn_apply_kernel in norms.hip)
__global__ __launch_bounds__(256) void gn_apply_hilo_kernel(const half_t* __restrict__ Xh, const half_t* __restrict__ Xl,
                                                            int ldx, half_t* __restrict__ Y, int ldy, int HW, int C,
                                                            int groups, const float* __restrict__ stats,
                                                            const half_t* __restrict__ gamma,
                                                            const half_t* __restrict__ beta, int silu) {
  const int b = blockIdx.y, chunk = blockIdx.x, nch = gridDim.x;
  const int cpg = C / groups;
  const int per = (HW + nch - 1) / nch;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  const int C8 = C >> 3;
  for (int item = threadIdx.x; item < C8 * 4; item += 256) {      // 4 pixel lanes per piece
    const int piece = item % C8, pl = item / C8;
    const int c0 = piece * 8;
    const half8_t gv = ld_half8(gamma + c0), bv = ld_half8(beta + c0);
    float a[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float* st = stats + ((size_t)b * groups + (c0 + j) / cpg) * 2;
      a[j] = st[1] * (float)gv[j];
      sh[j] = (float)bv[j] - st[0] * a[j];
    }
    for (int p = p0 + pl; p < p1; p += 4) {
      const size_t off = ((size_t)b * HW + p) * ldx + c0;
      const half8_t xh = ld_half8(Xh + off), xl = ld_half8(Xl + off);
      half8_t o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = fmaf((float)xh[j] + (float)xl[j], a[j], sh[j]);
        if (silu) v = silu_f(v);
        o[j] = (half_t)v;
      }
      st_half8(Y + ((size_t)b * HW + p) * ldy + c0, o);
    }
  }
}

// one wave per row, NQ 16-byte pieces per lane (C <= NQ * 512)
template <int NQ>
__global__ __launch_bounds__(256) void ln_fwd_hilo_kernel(const half_t* __restrict__ Xh, const half_t* __restrict__ Xl,
                                                          int ldx, half_t* __restrict__ Y, int ldy, int M, int C,
                                                          const half_t* __restrict__ gamma,
                                                          const half_t* __restrict__ beta, float eps,
                                                          float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int C8 = C >> 3;
  float v[NQ][8];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int pc = lane + q * 64;
    half8_t xh = zero_half8(), xl = zero_half8();
    if (pc < C8) { xh = ld_half8(Xh + (size_t)row * ldx + pc * 8); xl = ld_half8(Xl + (size_t)row * ldx + pc * 8); }
#pragma unroll
    for (int j = 0; j < 8; ++j) { v[q][j] = (float)xh[j] + (float)xl[j]; s += v[q][j]; }
  }
  const float mean = wave_sum(s) / C;
  float s2 = 0.f;
#pragma unroll
  for (int q = 0; q < NQ; ++q)
    if (lane + q * 64 < C8)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[q][j] - mean; s2 += d * d; }
  const float rstd = rsqrtf(wave_sum(s2) / C + eps);
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int pc = lane + q * 64;
    if (pc < C8) {
      const half8_t gv = ld_half8(gamma + pc * 8), bv = ld_half8(beta + pc * 8);
      half8_t o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (half_t)((v[q][j] - mean) * rstd * (float)gv[j] + (float)bv[j]);
      st_half8(Y + (size_t)row * ldy + pc * 8, o);
    }
  }
  if (stats && lane == 0) { stats[(size_t)row * 2] = mean; stats[(size_t)row * 2 + 1] = rstd; }      // for skg_layernorm_bwd
}

}  // namespace

extern "C" int skg_groupnorm_apply_hilo(const void* X, const void* X_lo, int ldx, void* Y, int ldy, int rows, int HW, int C,
                                        int groups, const float* stats, const void* gamma, const void* beta, int silu,
                                        void* stream) {
  SKG_REQUIRE(X && X_lo && Y && stats && gamma && beta && rows > 0 && HW > 0 && groups > 0 && C % 8 == 0 && C % groups == 0);
  SKG_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C);
  SKG_REQUIRE(skg_aligned(X, 16) && skg_aligned(X_lo, 16) && skg_aligned(Y, 16) && skg_aligned(gamma, 16) && skg_aligned(beta, 16));
  int nch = HW / 32;
  nch = nch > 256 ? 256 : (nch < 1 ? 1 : nch);
  hipLaunchKernelGGL(gn_apply_hilo_kernel, dim3(nch, rows), dim3(256), 0, (hipStream_t)stream, (const half_t*)X,
                     (const half_t*)X_lo, ldx, (half_t*)Y, ldy, HW, C, groups, stats, (const half_t*)gamma,
                     (const half_t*)beta, silu);
  SKG_CHECK_LAUNCH("skg_groupnorm_apply_hilo");
  return SKG_OK;
}

extern "C" int skg_layernorm_fwd_hilo(const void* X, const void* X_lo, int ldx, void* Y, int ldy, int M, int C,
                                      const void* gamma, const void* beta, float eps, float* stats, void* stream) {
  SKG_REQUIRE(X && X_lo && Y && gamma && beta && M > 0 && C % 8 == 0 && C <= 2048);
  SKG_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && skg_aligned(X, 16) && skg_aligned(X_lo, 16) && skg_aligned(Y, 16) &&
              skg_aligned(gamma, 16) && skg_aligned(beta, 16));
  hipStream_t st = (hipStream_t)stream;
  const half_t *xh = (const half_t*)X, *xl = (const half_t*)X_lo, *g = (const half_t*)gamma, *b = (const half_t*)beta;
  if (C <= 512)
    hipLaunchKernelGGL((ln_fwd_hilo_kernel<1>), dim3(skg_cdiv(M, 4)), dim3(256), 0, st, xh, xl, ldx, (half_t*)Y, ldy, M, C, g, b, eps, stats);
  else if (C <= 1024)
    hipLaunchKernelGGL((ln_fwd_hilo_kernel<2>), dim3(skg_cdiv(M, 4)), dim3(256), 0, st, xh, xl, ldx, (half_t*)Y, ldy, M, C, g, b, eps, stats);
  else
    hipLaunchKernelGGL((ln_fwd_hilo_kernel<4>), dim3(skg_cdiv(M, 4)), dim3(256), 0, st, xh, xl, ldx, (half_t*)Y, ldy, M, C, g, b, eps, stats);
  SKG_CHECK_LAUNCH("skg_layernorm_fwd_hilo");
  return SKG_OK;
}

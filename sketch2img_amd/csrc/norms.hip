// GroupNorm(+SiLU) and LayerNorm, forward and backward-to-input, NHWC / token-major fp16.
// All of these are HBM-bound streaming kernels: 16-byte loads where the layout allows, fp32
// statistics, wave-shuffle + LDS reductions, deterministic two-stage partial sums (no atomics
// on global memory, so results do not depend on workgroup scheduling).
#include "common.h"

namespace {

constexpr int GN_MAX_CHUNKS = 128;
constexpr int GN_MAX_C = 4096;

// pixels per workgroup ~32: enough workgroups (2048 at 64x64 x 16 rows) to cover the chip several times
__host__ __device__ inline int gn_chunks(int HW) {
  int c = (HW + 31) / 32;
  return c > GN_MAX_CHUNKS ? GN_MAX_CHUNKS : (c < 1 ? 1 : c);
}

// ---- stage 1: per (row, pixel-chunk) partial sums of two per-element quantities, per group ----
// KIND 0 (forward stats):   q1 = x,            q2 = x*x
// KIND 1 (backward sums):   q1 = dyh*gamma,    q2 = dyh*gamma*xhat     (dyh = dy * silu'(y))
// Each thread owns fixed 8-channel pieces (16-byte loads) for a strided subset of the chunk's pixels.  A piece
// can straddle two groups (cpg = 10, 30 ...), so it keeps a (low group, high group) pair of partial sums; the
// fold into groups runs in a fixed order from LDS -> bitwise deterministic results.
template <int KIND>
__global__ __launch_bounds__(256) void gn_partial_kernel(
    const half_t* __restrict__ X, int ldx, const half_t* __restrict__ dY, int lddy, int HW, int C,
    int groups, const float* __restrict__ stats, const half_t* __restrict__ gamma,
    const half_t* __restrict__ beta, int silu, float* __restrict__ partial) {
  constexpr int NP = GN_MAX_C / 8 / 256;       // pieces per thread when C/8 > 256 (2)
  __shared__ float red[256 * NP][4];
  const int b = blockIdx.y, chunk = blockIdx.x, nch = gridDim.x;
  const int cpg = C / groups;
  const int per = (HW + nch - 1) / nch;
  const int p0 = chunk * per;
  const int p1 = min(HW, p0 + per);
  const int C8 = C >> 3;
  const int P = C8 <= 256 ? 256 / C8 : 1;       // pixel lanes
  const int T = C8 <= 256 ? P * C8 : 256;       // active threads
  const int tid = threadIdx.x;
  const int pl = C8 <= 256 ? tid / C8 : 0;
  const int np = C8 <= 256 ? 1 : (C8 + 255) / 256;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    if (k >= np) break;
    const int piece = C8 <= 256 ? tid - pl * C8 : tid + k * 256;
    float s1lo = 0.f, s2lo = 0.f, s1hi = 0.f, s2hi = 0.f;
    if (tid < T && piece < C8) {
      const int c0 = piece * 8;
      const int glo = c0 / cpg;
      const int nlo = min(8, (glo + 1) * cpg - c0);        // channels [c0, c0+nlo) belong to glo, rest to glo+1
      float mlo = 0.f, rlo = 0.f, mhi = 0.f, rhi = 0.f;
      half8_t gv = zero_half8(), bv = zero_half8();
      if (KIND == 1) {
        mlo = stats[((size_t)b * groups + glo) * 2]; rlo = stats[((size_t)b * groups + glo) * 2 + 1];
        if (nlo < 8) { mhi = stats[((size_t)b * groups + glo + 1) * 2]; rhi = stats[((size_t)b * groups + glo + 1) * 2 + 1]; }
        gv = ld_half8(gamma + c0);
        bv = ld_half8(beta + c0);
      }
      for (int p = p0 + pl; p < p1; p += P) {
        const half8_t xv = ld_half8(X + ((size_t)b * HW + p) * ldx + c0);
        half8_t dv = zero_half8();
        if (KIND == 1) dv = ld_half8(dY + ((size_t)b * HW + p) * lddy + c0);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const float x = (float)xv[j];
          float q1, q2;
          if (KIND == 0) {
            q1 = x; q2 = x * x;
          } else {
            const bool lo = j < nlo;
            const float xh = (x - (lo ? mlo : mhi)) * (lo ? rlo : rhi);
            float d = (float)dv[j];
            if (silu) d *= silu_grad_f(xh * (float)gv[j] + (float)bv[j]);
            d *= (float)gv[j];
            q1 = d; q2 = d * xh;
          }
          if (j < nlo) { s1lo += q1; s2lo += q2; } else { s1hi += q1; s2hi += q2; }
        }
      }
    }
    red[tid + k * 256][0] = s1lo; red[tid + k * 256][1] = s2lo;
    red[tid + k * 256][2] = s1hi; red[tid + k * 256][3] = s2hi;
  }
  __syncthreads();
  if (tid < groups) {
    // pieces overlapping group tid: [first, last]; for each, add the half that belongs to this group
    const int first = (tid * cpg) >> 3, last = ((tid + 1) * cpg - 1) >> 3;
    float t1 = 0.f, t2 = 0.f;
    for (int piece = first; piece <= last; ++piece) {
      const bool is_lo = (piece * 8) / cpg == tid;          // this group is the piece's low group
      const int o = is_lo ? 0 : 2;
      if (C8 <= 256) {
        for (int l = 0; l < P; ++l) { t1 += red[l * C8 + piece][o]; t2 += red[l * C8 + piece][o + 1]; }
      } else {
        const int slot = (piece & 255) + (piece >> 8) * 256;
        t1 += red[slot][o]; t2 += red[slot][o + 1];
      }
    }
    float* o2 = partial + (((size_t)b * nch + chunk) * groups + tid) * 2;
    o2[0] = t1;
    o2[1] = t2;
  }
}

// ---- stage 2: fold the chunk partials.  KIND 0 -> (mean, rstd); KIND 1 -> (m1, m2) -------------
template <int KIND>
__global__ __launch_bounds__(256) void gn_finalize_kernel(const float* __restrict__ partial, int nch, int groups,
                                                          float inv_n, float eps, float* __restrict__ out, int total) {
  // one 16-lane group per (row, group): lanes stride over the chunk partials, fixed-order xor reduction
  const int i = (blockIdx.x * 256 + threadIdx.x) >> 4;
  const int sub = threadIdx.x & 15;
  float s1 = 0.f, s2 = 0.f;
  if (i < total) {
    const int b = i / groups, grp = i - b * groups;
    for (int c = sub; c < nch; c += 16) {
      const float* q = partial + (((size_t)b * nch + c) * groups + grp) * 2;
      s1 += q[0]; s2 += q[1];
    }
  }
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) {
    s1 += __shfl_xor(s1, o, 64);
    s2 += __shfl_xor(s2, o, 64);
  }
  if (i >= total || sub != 0) return;
  if (KIND == 0) {
    const float mean = s1 * inv_n;
    const float var = fmaxf(s2 * inv_n - mean * mean, 0.f);
    out[(size_t)i * 2] = mean;
    out[(size_t)i * 2 + 1] = rsqrtf(var + eps);
  } else {
    out[(size_t)i * 2] = s1 * inv_n;
    out[(size_t)i * 2 + 1] = s2 * inv_n;
  }
}

// ---- apply (forward): y = act((x - mean) * rstd * gamma + beta) ------------------------------------
__global__ __launch_bounds__(256) void gn_apply_kernel(const half_t* __restrict__ X, int ldx,
                                                       half_t* __restrict__ Y, int ldy, int rows, int HW,
                                                       int C, int groups, const float* __restrict__ stats,
                                                       const half_t* __restrict__ gamma,
                                                       const half_t* __restrict__ beta, int silu) {
  const int C8 = C >> 3;
  const int cpg = C / groups;
  const size_t total = (size_t)rows * HW * C8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t m = i / C8;
    const int c0 = (int)(i - m * C8) * 8;
    const int b = (int)(m / HW);
    const half8_t xv = ld_half8(X + m * ldx + c0);
    const half8_t gv = ld_half8(gamma + c0);
    const half8_t bv = ld_half8(beta + c0);
    const float* st = stats + (size_t)b * groups * 2;
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int grp = (c0 + j) / cpg;
      float v = ((float)xv[j] - st[grp * 2]) * st[grp * 2 + 1] * (float)gv[j] + (float)bv[j];
      if (silu) v = silu_f(v);
      o[j] = (half_t)v;
    }
    st_half8(Y + m * ldy + c0, o);
  }
}

// ---- apply (backward): dx = rstd * (dyh*gamma - m1 - xhat*m2) + residual --------------------------
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(
    const half_t* __restrict__ X, int ldx, const half_t* __restrict__ dY, int lddy, half_t* __restrict__ dX,
    int lddx, const half_t* __restrict__ R, int ldr, int rows, int HW, int C, int groups,
    const float* __restrict__ stats, const float* __restrict__ sums, const half_t* __restrict__ gamma,
    const half_t* __restrict__ beta, int silu) {
  const int C8 = C >> 3;
  const int cpg = C / groups;
  const size_t total = (size_t)rows * HW * C8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t m = i / C8;
    const int c0 = (int)(i - m * C8) * 8;
    const int b = (int)(m / HW);
    const half8_t xv = ld_half8(X + m * ldx + c0);
    const half8_t dv = ld_half8(dY + m * lddy + c0);
    const half8_t gv = ld_half8(gamma + c0);
    const half8_t bv = ld_half8(beta + c0);
    half8_t rv = zero_half8();
    if (R) rv = ld_half8(R + m * ldr + c0);
    const float* st = stats + (size_t)b * groups * 2;
    const float* sm = sums + (size_t)b * groups * 2;
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int grp = (c0 + j) / cpg;
      const float rstd = st[grp * 2 + 1];
      const float xh = ((float)xv[j] - st[grp * 2]) * rstd;
      float d = (float)dv[j];
      if (silu) d *= silu_grad_f(xh * (float)gv[j] + (float)bv[j]);
      d *= (float)gv[j];
      o[j] = (half_t)(rstd * (d - sm[grp * 2] - xh * sm[grp * 2 + 1]) + (float)rv[j]);
    }
    st_half8(dX + m * lddx + c0, o);
  }
}

// ---- LayerNorm: one wave per row, values held in registers ----------------------------------------
constexpr int LN_MAXP = 4;   // 16-byte pieces per lane -> C <= 2048

__global__ __launch_bounds__(256) void ln_fwd_kernel(const half_t* __restrict__ X, int ldx, half_t* __restrict__ Y,
                                                     int ldy, int M, int C, const half_t* __restrict__ gamma,
                                                     const half_t* __restrict__ beta, float eps,
                                                     float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int C8 = C >> 3;
  float v[LN_MAXP][8];
  float s = 0.f;
#pragma unroll
  for (int q = 0; q < LN_MAXP; ++q) {
    const int pc = lane + q * 64;
    if (pc < C8) {
      const half8_t x = ld_half8(X + (size_t)row * ldx + pc * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[q][j] = (float)x[j]; s += v[q][j]; }
    }
  }
  const float mean = wave_sum(s) / C;
  float s2 = 0.f;
#pragma unroll
  for (int q = 0; q < LN_MAXP; ++q)
    if (lane + q * 64 < C8)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = v[q][j] - mean; s2 += d * d; }
  const float rstd = rsqrtf(wave_sum(s2) / C + eps);
#pragma unroll
  for (int q = 0; q < LN_MAXP; ++q) {
    const int pc = lane + q * 64;
    if (pc < C8) {
      const half8_t gv = ld_half8(gamma + pc * 8), bv = ld_half8(beta + pc * 8);
      half8_t o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (half_t)((v[q][j] - mean) * rstd * (float)gv[j] + (float)bv[j]);
      st_half8(Y + (size_t)row * ldy + pc * 8, o);
    }
  }
  if (stats && lane == 0) { stats[(size_t)row * 2] = mean; stats[(size_t)row * 2 + 1] = rstd; }
}

__global__ __launch_bounds__(256) void ln_bwd_kernel(const half_t* __restrict__ X, int ldx,
                                                     const half_t* __restrict__ dY, int lddy,
                                                     half_t* __restrict__ dX, int lddx,
                                                     const half_t* __restrict__ R, int ldr, int M, int C,
                                                     const half_t* __restrict__ gamma,
                                                     const float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= M) return;
  const int C8 = C >> 3;
  const float mean = stats[(size_t)row * 2], rstd = stats[(size_t)row * 2 + 1];
  float xh[LN_MAXP][8], dg[LN_MAXP][8];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int q = 0; q < LN_MAXP; ++q) {
    const int pc = lane + q * 64;
    if (pc < C8) {
      const half8_t x = ld_half8(X + (size_t)row * ldx + pc * 8);
      const half8_t d = ld_half8(dY + (size_t)row * lddy + pc * 8);
      const half8_t gv = ld_half8(gamma + pc * 8);
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        xh[q][j] = ((float)x[j] - mean) * rstd;
        dg[q][j] = (float)d[j] * (float)gv[j];
        s1 += dg[q][j];
        s2 += dg[q][j] * xh[q][j];
      }
    }
  }
  const float m1 = wave_sum(s1) / C, m2 = wave_sum(s2) / C;
#pragma unroll
  for (int q = 0; q < LN_MAXP; ++q) {
    const int pc = lane + q * 64;
    if (pc < C8) {
      half8_t rv = zero_half8();
      if (R) rv = ld_half8(R + (size_t)row * ldr + pc * 8);
      half8_t o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (half_t)(rstd * (dg[q][j] - m1 - xh[q][j] * m2) + (float)rv[j]);
      st_half8(dX + (size_t)row * lddx + pc * 8, o);
    }
  }
}

inline int ew_grid(size_t total_items) {
  size_t b = (total_items + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" size_t skg_groupnorm_scratch_floats(int rows, int groups) {
  return (size_t)rows * GN_MAX_CHUNKS * groups * 2 + (size_t)rows * groups * 2;
}

extern "C" int skg_groupnorm_stats(const void* X, int ldx, int rows, int HW, int C, int groups, float eps,
                                   float* stats, float* partial, void* stream) {
  SKG_REQUIRE(X && stats && partial && rows > 0 && HW > 0 && groups > 0 && groups <= 64);
  SKG_REQUIRE(C % 8 == 0 && C % groups == 0 && (C / groups) % 2 == 0 && (C / groups) >= 4 && ldx % 8 == 0 && ldx >= C &&
              C <= GN_MAX_C && skg_aligned(X, 16));
  hipStream_t st = (hipStream_t)stream;
  const int nch = gn_chunks(HW);
  hipLaunchKernelGGL((gn_partial_kernel<0>), dim3(nch, rows), dim3(256), 0, st, (const half_t*)X, ldx,
                     (const half_t*)nullptr, 0, HW, C, groups, (const float*)nullptr, (const half_t*)nullptr,
                     (const half_t*)nullptr, 0, partial);
  const int total = rows * groups;
  hipLaunchKernelGGL((gn_finalize_kernel<0>), dim3(skg_cdiv(total * 16, 256)), dim3(256), 0, st, partial, nch,
                     groups, 1.f / ((float)HW * (C / groups)), eps, stats, total);
  SKG_CHECK_LAUNCH("skg_groupnorm_stats");
  return SKG_OK;
}

extern "C" int skg_groupnorm_apply(const void* X, int ldx, void* Y, int ldy, int rows, int HW, int C,
                                   int groups, const float* stats, const void* gamma, const void* beta,
                                   int silu, void* stream) {
  SKG_REQUIRE(X && Y && stats && gamma && beta && rows > 0 && HW > 0);
  SKG_REQUIRE(C % 8 == 0 && C % groups == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C);
  SKG_REQUIRE(skg_aligned(X, 16) && skg_aligned(Y, 16) && skg_aligned(gamma, 16) && skg_aligned(beta, 16));
  const size_t total = (size_t)rows * HW * (C / 8);
  hipLaunchKernelGGL(gn_apply_kernel, dim3(ew_grid(total)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)X, ldx, (half_t*)Y, ldy, rows, HW, C, groups, stats, (const half_t*)gamma,
                     (const half_t*)beta, silu);
  SKG_CHECK_LAUNCH("skg_groupnorm_apply");
  return SKG_OK;
}

extern "C" int skg_groupnorm_bwd(const void* X, int ldx, const void* dY, int lddy, void* dX, int lddx,
                                 const void* residual, int ldr, int rows, int HW, int C, int groups,
                                 const float* stats, const void* gamma, const void* beta, int silu,
                                 float* partial, void* stream) {
  SKG_REQUIRE(X && dY && dX && stats && gamma && beta && partial && rows > 0 && HW > 0 && groups <= 64);
  SKG_REQUIRE(C % 8 == 0 && C % groups == 0 && (C / groups) % 2 == 0 && (C / groups) >= 4 && C <= GN_MAX_C);
  SKG_REQUIRE(ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && (!residual || ldr % 8 == 0));
  SKG_REQUIRE(skg_aligned(X, 16) && skg_aligned(dY, 16) && skg_aligned(dX, 16) && skg_aligned(gamma, 16) &&
              skg_aligned(beta, 16) && (!residual || skg_aligned(residual, 16)));
  hipStream_t st = (hipStream_t)stream;
  const int nch = gn_chunks(HW);
  float* sums = partial + (size_t)rows * GN_MAX_CHUNKS * groups * 2;
  hipLaunchKernelGGL((gn_partial_kernel<1>), dim3(nch, rows), dim3(256), 0, st, (const half_t*)X, ldx,
                     (const half_t*)dY, lddy, HW, C, groups, stats, (const half_t*)gamma, (const half_t*)beta,
                     silu, partial);
  const int total = rows * groups;
  hipLaunchKernelGGL((gn_finalize_kernel<1>), dim3(skg_cdiv(total * 16, 256)), dim3(256), 0, st, partial, nch,
                     groups, 1.f / ((float)HW * (C / groups)), 0.f, sums, total);
  const size_t items = (size_t)rows * HW * (C / 8);
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(ew_grid(items)), dim3(256), 0, st, (const half_t*)X, ldx,
                     (const half_t*)dY, lddy, (half_t*)dX, lddx, (const half_t*)residual, ldr, rows, HW, C,
                     groups, stats, sums, (const half_t*)gamma, (const half_t*)beta, silu);
  SKG_CHECK_LAUNCH("skg_groupnorm_bwd");
  return SKG_OK;
}

extern "C" int skg_layernorm_fwd(const void* X, int ldx, void* Y, int ldy, int M, int C, const void* gamma,
                                 const void* beta, float eps, float* stats, void* stream) {
  SKG_REQUIRE(X && Y && gamma && beta && M > 0 && C % 8 == 0 && C <= LN_MAXP * 64 * 8);
  SKG_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && skg_aligned(X, 16) && skg_aligned(Y, 16) &&
              skg_aligned(gamma, 16) && skg_aligned(beta, 16));
  hipLaunchKernelGGL(ln_fwd_kernel, dim3(skg_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, (const half_t*)X,
                     ldx, (half_t*)Y, ldy, M, C, (const half_t*)gamma, (const half_t*)beta, eps, stats);
  SKG_CHECK_LAUNCH("skg_layernorm_fwd");
  return SKG_OK;
}

extern "C" int skg_layernorm_bwd(const void* X, int ldx, const void* dY, int lddy, void* dX, int lddx,
                                 const void* residual, int ldr, int M, int C, const void* gamma,
                                 const float* stats, void* stream) {
  SKG_REQUIRE(X && dY && dX && gamma && stats && M > 0 && C % 8 == 0 && C <= LN_MAXP * 64 * 8);
  SKG_REQUIRE(ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && (!residual || ldr % 8 == 0));
  SKG_REQUIRE(skg_aligned(X, 16) && skg_aligned(dY, 16) && skg_aligned(dX, 16) && skg_aligned(gamma, 16) &&
              (!residual || skg_aligned(residual, 16)));
  hipLaunchKernelGGL(ln_bwd_kernel, dim3(skg_cdiv(M, 4)), dim3(256), 0, (hipStream_t)stream, (const half_t*)X,
                     ldx, (const half_t*)dY, lddy, (half_t*)dX, lddx, (const half_t*)residual, ldr, M, C,
                     (const half_t*)gamma, stats);
  SKG_CHECK_LAUNCH("skg_layernorm_bwd");
  return SKG_OK;
}

// GroupNorm(+SiLU) and LayerNorm, forward and backward-to-input, NHWC / token-major fp16.
// All of these are HBM-bound streaming kernels: 16-byte loads where the layout allows, fp32
// statistics, wave-shuffle + LDS reductions, deterministic two-stage partial sums (no atomics
// on global memory, so results do not depend on workgroup scheduling).
#include "common.h"
#include "gemm_params.h"
#include <type_traits>

namespace {

constexpr int GN_MAX_CHUNKS = 128;
constexpr int GN_MAX_C = 4096;

// pixels per workgroup ~48: enough workgroups (1376 at 64x64 x 16 rows) to cover the chip several times
__host__ __device__ inline int gn_chunks(int HW) {
  int c = (HW + 47) / 48;                  // ~48 pixels per workgroup = 8 per thread at C = 320 (6 pixel lanes)
  return c > GN_MAX_CHUNKS ? GN_MAX_CHUNKS : (c < 1 ? 1 : c);
}

// apply kernels: each thread should see >= 8 pixels (two unrolled iterations of four) of its channel piece
__host__ __device__ inline int gn_apply_chunks(int HW, int C) {
  const int C8 = C >> 3;
  const int P = C8 <= 256 ? 256 / C8 : 1;
  int c = HW / (8 * P);
  return c > 512 ? 512 : (c < 1 ? 1 : c);
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
      auto accum = [&](const half8_t& xv, const half8_t& dv) {
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
      };
      // four pixels per iteration: the loads of all four are issued before the first is consumed (see gn_apply_kernel)
      const half_t* xp = X + ((size_t)b * HW) * ldx + c0;
      const half_t* dp = KIND == 1 ? dY + ((size_t)b * HW) * lddy + c0 : nullptr;
      int p = p0 + pl;
      for (; p + 3 * P < p1; p += 4 * P) {
        half8_t xv[4], dv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          xv[u] = ld_half8(xp + (size_t)(p + u * P) * ldx);
          dv[u] = KIND == 1 ? ld_half8(dp + (size_t)(p + u * P) * lddy) : zero_half8();
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) accum(xv[u], dv[u]);
      }
      for (; p < p1; p += P)
        accum(ld_half8(xp + (size_t)p * ldx), KIND == 1 ? ld_half8(dp + (size_t)p * lddy) : zero_half8());
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

// ---- apply (forward): y = act(x * a + sh), a = rstd * gamma, sh = beta - mean * a -------------------------
// Same decomposition as stage 1: workgroup = (pixel chunk, row), thread = fixed 8-channel piece x strided pixels, so
// the per-channel scale / shift are loop invariant registers; FOUR pixels per iteration so four independent 16-byte
// loads are in flight per thread.  (One load in flight per thread caps a streaming kernel at ~4 TB/s on this chip -
// 32 waves x 64 lanes x 16 B per CU against ~2 us of latency; the first version, which also re-derived the group of
// every element, ran at 2.6 TB/s.)
// FOLD: the statistics are folded here from the stage-1 chunk partials (every workgroup of a row repeats the same
// fixed-order fold into LDS; chunk 0 also publishes (mean, rstd) for the backward pass) - this removes the separate
// finalize launch (4.7 us of an otherwise ~15-40 us GroupNorm).
struct GnSrc { const float* partialB; int split, unused, ratioA, ratioB; };
// HILO (accuracy mode, include/skg.h): the input is the PAIR X + Xl of fp16 tensors with one pitch (what the hi / lo epilogues
// of gemm2.hip / gemm8.hip write); the statistics are those of the hi part (the producer's epilogue sums, or gn_partial_kernel
// on X): the mean of >= 640 rounding errors of relative size 2^-12 is far below fp32 resolution.
template <bool FOLD, bool HILO = false>
__global__ __launch_bounds__(256) void gn_apply_kernel(const half_t* __restrict__ X, const half_t* __restrict__ Xl, int ldx,
                                                       half_t* __restrict__ Y, half_t* __restrict__ Yl, int ldy, int HW, int C, int groups,
                                                       const float* __restrict__ stats,
                                                       const half_t* __restrict__ gamma,
                                                       const half_t* __restrict__ beta, int silu,
                                                       const float* __restrict__ partial, int pch, float inv_n,
                                                       float eps, float* __restrict__ stats_out,
                                                       GnSrc src = GnSrc{nullptr, 0, 0, 1, 1}) {
  const int b = blockIdx.y, chunk = blockIdx.x, nch = gridDim.x;
  __shared__ float st_s[64][2];
  if (FOLD) {
    // 8 lanes per group, lanes stride over the pch chunk partials, xor-shuffle fold (fixed order).  `src` (GroupNorm of a
    // CONCATENATION whose halves were written by two producers): groups below src.split come from `partial`, the others
    // from src.partialB, and every output group adds ratioA / ratioB adjacent (narrower) source groups.
    const int split = src.partialB ? src.split : groups;
    const int sgA = split * src.ratioA, sgB = (groups - split) * src.ratioB;
    for (int g0 = 0; g0 < groups; g0 += 32) {
      const int grp = g0 + (threadIdx.x >> 3), sub = threadIdx.x & 7;
      float s1 = 0.f, s2 = 0.f;
      if (grp < groups) {
        const bool inA = grp < split;
        const float* base = inA ? partial : src.partialB;
        const int sg = inA ? sgA : sgB, r = inA ? src.ratioA : src.ratioB, gs = (inA ? grp : grp - split) * r;
        for (int c = sub; c < pch; c += 8) {
          const float* q = base + (((size_t)b * pch + c) * sg + gs) * 2;
          for (int k = 0; k < r; ++k) { s1 += q[2 * k]; s2 += q[2 * k + 1]; }
        }
      }
#pragma unroll
      for (int o = 4; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
      if (grp < groups && sub == 0) {
        const float mean = s1 * inv_n;
        const float var = fmaxf(s2 * inv_n - mean * mean, 0.f);
        const float rstd = rsqrtf(var + eps);
        st_s[grp][0] = mean; st_s[grp][1] = rstd;
        if (chunk == 0) { stats_out[((size_t)b * groups + grp) * 2] = mean; stats_out[((size_t)b * groups + grp) * 2 + 1] = rstd; }
      }
    }
    __syncthreads();
  }
  const int cpg = C / groups;
  const int per = (HW + nch - 1) / nch;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  const int C8 = C >> 3;
  const int P = C8 <= 256 ? 256 / C8 : 1;       // pixel lanes
  const int T = C8 <= 256 ? P * C8 : 256;       // active threads
  const int tid = threadIdx.x;
  const int pl = C8 <= 256 ? tid / C8 : 0;
  const int np = C8 <= 256 ? 1 : (C8 + 255) / 256;
  for (int k = 0; k < np; ++k) {
    const int piece = C8 <= 256 ? tid - pl * C8 : tid + k * 256;
    if (tid >= T || piece >= C8) continue;
    const int c0 = piece * 8;
    const int glo = c0 / cpg;
    const int nlo = min(8, (glo + 1) * cpg - c0);        // channels [c0, c0+nlo) belong to glo, the rest to glo+1
    const float* st = FOLD ? &st_s[glo][0] : stats + ((size_t)b * groups + glo) * 2;
    const float mlo = st[0], rlo = st[1];
    const float mhi = nlo < 8 ? st[2] : 0.f, rhi = nlo < 8 ? st[3] : 0.f;
    const half8_t gv = ld_half8(gamma + c0), bv = ld_half8(beta + c0);
    float a[8], sh[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const bool lo = j < nlo;
      a[j] = (lo ? rlo : rhi) * (float)gv[j];
      sh[j] = (float)bv[j] - (lo ? mlo : mhi) * a[j];
    }
    const half_t* xp = X + ((size_t)b * HW) * ldx + c0;
    const half_t* lp = HILO ? Xl + ((size_t)b * HW) * ldx + c0 : nullptr;
    half_t* yp = Y + ((size_t)b * HW) * ldy + c0;
    half_t* ylp = (HILO && Yl) ? Yl + ((size_t)b * HW) * ldy + c0 : nullptr;      // HILO: optional PAIR output (lo = fp16(v - fp16(v)))
    int p = p0 + pl;
    constexpr int U = HILO ? 2 : 4;      // independent pixels per iteration: four 16-byte loads in flight either way
    for (; p + (U - 1) * P < p1; p += U * P) {
      half8_t xv[U], lv[U];
#pragma unroll
      for (int u = 0; u < U; ++u) {
        xv[u] = ld_half8(xp + (size_t)(p + u * P) * ldx);
        if (HILO) lv[u] = ld_half8(lp + (size_t)(p + u * P) * ldx);
      }
#pragma unroll
      for (int u = 0; u < U; ++u) {
        half8_t o, ol;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float v = fmaf(HILO ? (float)xv[u][j] + (float)lv[u][j] : (float)xv[u][j], a[j], sh[j]);
          if (silu) v = silu_f(v);
          o[j] = (half_t)v;
          if (HILO) ol[j] = (half_t)(v - (float)o[j]);
        }
        st_half8(yp + (size_t)(p + u * P) * ldy, o);
        if (HILO) { if (ylp) st_half8(ylp + (size_t)(p + u * P) * ldy, ol); }
      }
    }
    for (; p < p1; p += P) {
      const half8_t xv = ld_half8(xp + (size_t)p * ldx);
      const half8_t lv = HILO ? ld_half8(lp + (size_t)p * ldx) : zero_half8();
      half8_t o, ol;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float v = fmaf(HILO ? (float)xv[j] + (float)lv[j] : (float)xv[j], a[j], sh[j]);
        if (silu) v = silu_f(v);
        o[j] = (half_t)v;
        if (HILO) ol[j] = (half_t)(v - (float)o[j]);
      }
      st_half8(yp + (size_t)p * ldy, o);
      if (HILO) { if (ylp) st_half8(ylp + (size_t)p * ldy, ol); }
    }
  }
}

// ---- apply (backward): dx = rstd * (dyh*gamma - m1 - xhat*m2) + residual --------------------------
// same decomposition and unrolling as gn_apply_kernel (two pixels per iteration: three streams per pixel)
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(
    const half_t* __restrict__ X, int ldx, const half_t* __restrict__ dY, int lddy, half_t* __restrict__ dX,
    int lddx, const half_t* __restrict__ R, int ldr, int HW, int C, int groups,
    const float* __restrict__ stats, const float* __restrict__ partial, int pch, float inv_n, const half_t* __restrict__ gamma,
    const half_t* __restrict__ beta, int silu) {
  const int b = blockIdx.y, chunk = blockIdx.x, nch = gridDim.x;
  // round 6: the fold of the stage-1 chunk partials (m1 = mean(d), m2 = mean(d * xhat) per group) happens here - every workgroup of a
  // row repeats gn_finalize_kernel<1>'s fixed-order fold (16 lanes per group striding over the chunks, xor tree 8, 4, 2, 1: the same
  // bits) into LDS; one launch per GroupNorm backward less (598 per config-2 batch)
  __shared__ float sm_s[64][2];
  for (int g0 = 0; g0 < groups; g0 += 16) {
    const int grp = g0 + (threadIdx.x >> 4), sub = threadIdx.x & 15;
    float s1 = 0.f, s2 = 0.f;
    if (grp < groups)
      for (int c = sub; c < pch; c += 16) {
        const float* q = partial + (((size_t)b * pch + c) * groups + grp) * 2;
        s1 += q[0]; s2 += q[1];
      }
#pragma unroll
    for (int o = 8; o > 0; o >>= 1) { s1 += __shfl_xor(s1, o, 64); s2 += __shfl_xor(s2, o, 64); }
    if (grp < groups && sub == 0) { sm_s[grp][0] = s1 * inv_n; sm_s[grp][1] = s2 * inv_n; }
  }
  __syncthreads();
  const int cpg = C / groups;
  const int per = (HW + nch - 1) / nch;
  const int p0 = chunk * per, p1 = min(HW, p0 + per);
  const int C8 = C >> 3;
  const int P = C8 <= 256 ? 256 / C8 : 1;
  const int T = C8 <= 256 ? P * C8 : 256;
  const int tid = threadIdx.x;
  const int pl = C8 <= 256 ? tid / C8 : 0;
  const int np = C8 <= 256 ? 1 : (C8 + 255) / 256;
  for (int k = 0; k < np; ++k) {
    const int piece = C8 <= 256 ? tid - pl * C8 : tid + k * 256;
    if (tid >= T || piece >= C8) continue;
    const int c0 = piece * 8;
    const int glo = c0 / cpg;
    const int nlo = min(8, (glo + 1) * cpg - c0);
    const float* st = stats + ((size_t)b * groups + glo) * 2;
    const bool two = nlo < 8;
    const float mlo = st[0], rlo = st[1], m1lo = sm_s[glo][0], m2lo = sm_s[glo][1];
    const float mhi = two ? st[2] : 0.f, rhi = two ? st[3] : 0.f, m1hi = two ? sm_s[glo + 1][0] : 0.f, m2hi = two ? sm_s[glo + 1][1] : 0.f;
    const half8_t gv = ld_half8(gamma + c0), bv = ld_half8(beta + c0);
    const size_t row0 = (size_t)b * HW;
    auto one = [&](const half8_t& xv, const half8_t& dv, const half8_t& rv, size_t m) {
      half8_t o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool lo = j < nlo;
        const float rstd = lo ? rlo : rhi;
        const float xh = ((float)xv[j] - (lo ? mlo : mhi)) * rstd;
        float d = (float)dv[j];
        if (silu) d *= silu_grad_f(xh * (float)gv[j] + (float)bv[j]);
        d *= (float)gv[j];
        o[j] = (half_t)(rstd * (d - (lo ? m1lo : m1hi) - xh * (lo ? m2lo : m2hi)) + (float)rv[j]);
      }
      st_half8(dX + m * lddx + c0, o);
    };
    int p = p0 + pl;
    for (; p + P < p1; p += 2 * P) {
      const size_t ma = row0 + p, mb = row0 + p + P;
      const half8_t xa = ld_half8(X + ma * ldx + c0), xb = ld_half8(X + mb * ldx + c0);
      const half8_t da = ld_half8(dY + ma * lddy + c0), db = ld_half8(dY + mb * lddy + c0);
      const half8_t ra = R ? ld_half8(R + ma * ldr + c0) : zero_half8();
      const half8_t rb = R ? ld_half8(R + mb * ldr + c0) : zero_half8();
      one(xa, da, ra, ma);
      one(xb, db, rb, mb);
    }
    for (; p < p1; p += P) {
      const size_t m = row0 + p;
      one(ld_half8(X + m * ldx + c0), ld_half8(dY + m * lddy + c0), R ? ld_half8(R + m * ldr + c0) : zero_half8(), m);
    }
  }
}

// ---- LayerNorm: one wave per row, values held in registers ----------------------------------------
constexpr int LN_MAXP = 4;   // 16-byte pieces per lane -> C <= 2048

// NQ 16-byte pieces per lane cover a row (C <= NQ*512), RW rows per wave are processed together so that NQ*RW = 4
// independent loads are in flight per lane (one load per lane caps a streaming kernel at ~4 TB/s, see gn_apply_kernel)
template <int NQ, int RW, bool HILO = false>      // HILO: the input is the pair X + Xl (accuracy mode)
__global__ __launch_bounds__(256) void ln_fwd_kernel(const half_t* __restrict__ X, const half_t* __restrict__ Xl, int ldx,
                                                     half_t* __restrict__ Y, int ldy, int M, int C, const half_t* __restrict__ gamma,
                                                     const half_t* __restrict__ beta, float eps,
                                                     float* __restrict__ stats) {
  const int lane = threadIdx.x & 63;
  const int row0 = (blockIdx.x * 4 + (threadIdx.x >> 6)) * RW;
  if (row0 >= M) return;
  const int C8 = C >> 3;
  float v[RW][NQ][8];
  float s[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    s[r] = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int pc = lane + q * 64;
      half8_t x = zero_half8(), xl = zero_half8();
      if (pc < C8 && row0 + r < M) {
        x = ld_half8(X + (size_t)(row0 + r) * ldx + pc * 8);
        if (HILO) xl = ld_half8(Xl + (size_t)(row0 + r) * ldx + pc * 8);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) { v[r][q][j] = HILO ? (float)x[j] + (float)xl[j] : (float)x[j]; s[r] += v[r][q][j]; }
    }
  }
  float mean[RW], rstd[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) mean[r] = wave_sum(s[r]) / C;
#pragma unroll
  for (int r = 0; r < RW; ++r) {
    float s2 = 0.f;
#pragma unroll
    for (int q = 0; q < NQ; ++q)
      if (lane + q * 64 < C8)
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[r][q][j] - mean[r]; s2 += d * d; }
    rstd[r] = rsqrtf(wave_sum(s2) / C + eps);
  }
#pragma unroll
  for (int q = 0; q < NQ; ++q) {
    const int pc = lane + q * 64;
    if (pc < C8) {
      const half8_t gv = ld_half8(gamma + pc * 8), bv = ld_half8(beta + pc * 8);
#pragma unroll
      for (int r = 0; r < RW; ++r) {
        if (row0 + r >= M) break;
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j)
          o[j] = (half_t)((v[r][q][j] - mean[r]) * rstd[r] * (float)gv[j] + (float)bv[j]);
        st_half8(Y + (size_t)(row0 + r) * ldy + pc * 8, o);
      }
    }
  }
  if (stats && lane == 0) {
#pragma unroll
    for (int r = 0; r < RW; ++r)
      if (row0 + r < M) { stats[(size_t)(row0 + r) * 2] = mean[r]; stats[(size_t)(row0 + r) * 2 + 1] = rstd[r]; }
  }
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
  half8_t rres[LN_MAXP];          // the residual is fetched with the other two streams, not after the reductions
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int q = 0; q < LN_MAXP; ++q) {
    const int pc = lane + q * 64;
    rres[q] = zero_half8();
    if (pc < C8) {
      const half8_t x = ld_half8(X + (size_t)row * ldx + pc * 8);
      const half8_t d = ld_half8(dY + (size_t)row * lddy + pc * 8);
      if (R) rres[q] = ld_half8(R + (size_t)row * ldr + pc * 8);
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
      half8_t o;
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (half_t)(rstd * (dg[q][j] - m1 - xh[q][j] * m2) + (float)rres[q][j]);
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

static int gn_apply_impl(const void* X, const void* Xl, int ldx, void* Y, int ldy, int rows, int HW, int C, int groups,
                         const float* stats, const void* gamma, const void* beta, int silu, void* stream) {
  SKG_REQUIRE(X && Y && stats && gamma && beta && rows > 0 && HW > 0);
  SKG_REQUIRE(C % 8 == 0 && C % groups == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C);
  SKG_REQUIRE((C / groups) >= 4 && C <= GN_MAX_C);      // an 8-channel piece spans at most two groups
  SKG_REQUIRE(skg_aligned(X, 16) && skg_aligned(Y, 16) && skg_aligned(gamma, 16) && skg_aligned(beta, 16) && skg_aligned(Xl, 16));
#define SKG_GN_APPLY(H)                                                                                                       \
  hipLaunchKernelGGL((gn_apply_kernel<false, H>), dim3(gn_apply_chunks(HW, C), rows), dim3(256), 0, (hipStream_t)stream,      \
                     (const half_t*)X, (const half_t*)Xl, ldx, (half_t*)Y, (half_t*)nullptr, ldy, HW, C, groups, stats, (const half_t*)gamma,   \
                     (const half_t*)beta, silu, (const float*)nullptr, 0, 0.f, 0.f, (float*)nullptr)
  if (Xl) SKG_GN_APPLY(true); else SKG_GN_APPLY(false);
#undef SKG_GN_APPLY
  SKG_CHECK_LAUNCH("skg_groupnorm_apply");
  return SKG_OK;
}

extern "C" int skg_groupnorm_apply(const void* X, int ldx, void* Y, int ldy, int rows, int HW, int C,
                                   int groups, const float* stats, const void* gamma, const void* beta,
                                   int silu, void* stream) {
  return gn_apply_impl(X, nullptr, ldx, Y, ldy, rows, HW, C, groups, stats, gamma, beta, silu, stream);
}

extern "C" int skg_groupnorm_apply_hilo(const void* X, const void* X_lo, int ldx, void* Y, int ldy, int rows, int HW, int C,
                                        int groups, const float* stats, const void* gamma, const void* beta, int silu,
                                        void* stream) {
  SKG_REQUIRE(X_lo);
  return gn_apply_impl(X, X_lo, ldx, Y, ldy, rows, HW, C, groups, stats, gamma, beta, silu, stream);
}

// Small maps (16x16 / 8x8 levels): ONE workgroup per (row, group) keeps the group's whole slice in registers - X is
// read once, mean and the CENTRED variance come from two block reductions, the statistics are published for the
// backward pass.  The chunked path is launch-latency bound there (two launches, 14-20 us for 2.6-10 MB).
template <int NP, bool HILO = false>   // 16-byte pieces per thread; HILO: the input is the pair X + Xl (see gn_apply_kernel)
__global__ __launch_bounds__(256) void gn_small_kernel(const half_t* __restrict__ X, const half_t* __restrict__ Xl, int ldx,
                                                       half_t* __restrict__ Y, half_t* __restrict__ Yl, int ldy, int HW, int C, int groups,
                                                       const half_t* __restrict__ gamma, const half_t* __restrict__ beta,
                                                       int silu, float eps, float* __restrict__ stats) {
  __shared__ float red[8];
  const int g = blockIdx.x, row = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / groups, ppp = cpg >> 3;              // channels per group, pieces per pixel
  const int npieces = HW * ppp;
  const half_t* xb = X + (size_t)row * HW * ldx + g * cpg;
  half_t* yb = Y + (size_t)row * HW * ldy + g * cpg;
  typedef float float8_t __attribute__((ext_vector_type(8)));
  typedef typename std::conditional<HILO, float8_t, half8_t>::type val8_t;      // the pair's sum needs fp32 registers
  val8_t v[NP];
  int off[NP];                                             // pixel * ld is recomputed for the store; keep (px, pc)
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int pi = tid + k * 256;
    const int px = pi / ppp, pc = pi - px * ppp;
    off[k] = pi < npieces ? (px << 8) | pc : -1;           // ppp <= 255
    const half8_t h = pi < npieces ? ld_half8(xb + (size_t)px * ldx + pc * 8) : zero_half8();
    const half8_t l = (HILO && pi < npieces) ? ld_half8(Xl + (size_t)row * HW * ldx + g * cpg + (size_t)px * ldx + pc * 8) : zero_half8();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      if constexpr (HILO) v[k][j] = (float)h[j] + (float)l[j];
      else v[k][j] = h[j];
      s += (float)v[k][j];
    }
  }
  const float inv_n = 1.f / ((float)HW * cpg);
  const float mean = block_sum<256>(s, red) * inv_n;
  float s2 = 0.f;
#pragma unroll
  for (int k = 0; k < NP; ++k)
    if (off[k] >= 0)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = (float)v[k][j] - mean; s2 += d * d; }
  const float rstd = rsqrtf(block_sum<256>(s2, red) * inv_n + eps);
  if (tid == 0) { stats[((size_t)row * groups + g) * 2] = mean; stats[((size_t)row * groups + g) * 2 + 1] = rstd; }
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    if (off[k] < 0) continue;
    const int px = off[k] >> 8, pc = off[k] & 255;
    const half8_t gv = ld_half8(gamma + g * cpg + pc * 8), bv = ld_half8(beta + g * cpg + pc * 8);
    half8_t o, ol;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float y = ((float)v[k][j] - mean) * rstd * (float)gv[j] + (float)bv[j];
      if (silu) y = silu_f(y);
      o[j] = (half_t)y;
      if (HILO) ol[j] = (half_t)(y - (float)o[j]);
    }
    st_half8(yb + (size_t)px * ldy + pc * 8, o);
    if (HILO) { if (Yl) st_half8(Yl + (size_t)row * HW * ldy + g * cpg + (size_t)px * ldy + pc * 8, ol); }
  }
}

// The same for a GroupNorm whose consumer is a Winograd F(2x2, 3x3) convolution (wino.hip, round 6): the normalised fp16 slice goes to
// LDS instead of memory and the workgroup writes its INPUT TRANSFORM V = B^T n B - per 4 x 4 tile (stride 2, zero halo) and 8-channel
// piece: the adds in fp32 on the fp16-rounded activations, one rounding, component c = 4 i + j at columns [c C, (c + 1) C) of
// V [rows * IH/2 * IW/2][16 C] - bit-identical to gn_small_kernel followed by wino_in_kernel, one launch and one [M, C] round trip less.
template <int NP>
__global__ __launch_bounds__(256) void gn_small_wino_kernel(const half_t* __restrict__ X, int ldx, half_t* __restrict__ V, int IH, int IW, int C,
                                                            int groups, const half_t* __restrict__ gamma, const half_t* __restrict__ beta,
                                                            int silu, float eps, float* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) half_t gsl[];      // [HW][cpg]
  __shared__ float red[8];
  const int g = blockIdx.x, row = blockIdx.y, tid = threadIdx.x;
  const int HW = IH * IW;
  const int cpg = C / groups, ppp = cpg >> 3;
  const int npieces = HW * ppp;
  const half_t* xb = X + (size_t)row * HW * ldx + g * cpg;
  half8_t v[NP];
  int off[NP];
  float s = 0.f;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int pi = tid + k * 256;
    const int px = pi / ppp, pc = pi - px * ppp;
    off[k] = pi < npieces ? (px << 8) | pc : -1;
    v[k] = pi < npieces ? ld_half8(xb + (size_t)px * ldx + pc * 8) : zero_half8();
#pragma unroll
    for (int j = 0; j < 8; ++j) s += (float)v[k][j];
  }
  const float inv_n = 1.f / ((float)HW * cpg);
  const float mean = block_sum<256>(s, red) * inv_n;
  float s2 = 0.f;
#pragma unroll
  for (int k = 0; k < NP; ++k)
    if (off[k] >= 0)
#pragma unroll
      for (int j = 0; j < 8; ++j) { const float d = (float)v[k][j] - mean; s2 += d * d; }
  const float rstd = rsqrtf(block_sum<256>(s2, red) * inv_n + eps);
  if (tid == 0) { stats[((size_t)row * groups + g) * 2] = mean; stats[((size_t)row * groups + g) * 2 + 1] = rstd; }
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    if (off[k] < 0) continue;
    const int px = off[k] >> 8, pc = off[k] & 255;
    const half8_t gv = ld_half8(gamma + g * cpg + pc * 8), bv = ld_half8(beta + g * cpg + pc * 8);
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float y = ((float)v[k][j] - mean) * rstd * (float)gv[j] + (float)bv[j];
      if (silu) y = silu_f(y);
      o[j] = (half_t)y;
    }
    st_half8(gsl + (size_t)px * cpg + pc * 8, o);
  }
  __syncthreads();
  const int TW = IW >> 1, tiles = (IH >> 1) * TW;
  for (int it = tid; it < tiles * ppp; it += 256) {
    const int tile = it / ppp, pc = it - tile * ppp;
    const int ti = tile / TW, tj = tile - ti * TW;
    float t[4][4][8];      // B^T d (rows transformed), per column
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int x = 2 * tj - 1 + b;
      float d[4][8];
#pragma unroll
      for (int a = 0; a < 4; ++a) {
        const int y = 2 * ti - 1 + a;
        half8_t q = zero_half8();
        if (x >= 0 && x < IW && y >= 0 && y < IH) q = ld_half8(gsl + (size_t)(y * IW + x) * cpg + pc * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) d[a][e] = (float)q[e];
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        t[0][b][e] = d[0][e] - d[2][e];
        t[1][b][e] = d[1][e] + d[2][e];
        t[2][b][e] = d[2][e] - d[1][e];
        t[3][b][e] = d[1][e] - d[3][e];
      }
    }
    half_t* out = V + ((size_t)row * tiles + tile) * (16 * (size_t)C) + g * cpg + pc * 8;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      half8_t v0, v1, v2, v3;
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        v0[e] = (half_t)(t[a][0][e] - t[a][2][e]);
        v1[e] = (half_t)(t[a][1][e] + t[a][2][e]);
        v2[e] = (half_t)(t[a][2][e] - t[a][1][e]);
        v3[e] = (half_t)(t[a][1][e] - t[a][3][e]);
      }
      st_half8(out + (size_t)(4 * a + 0) * C, v0);
      st_half8(out + (size_t)(4 * a + 1) * C, v1);
      st_half8(out + (size_t)(4 * a + 2) * C, v2);
      st_half8(out + (size_t)(4 * a + 3) * C, v3);
    }
  }
}

// backward of the same: dx = rstd * (d - mean(d) - xhat * mean(d * xhat)) + residual,  d = dy * silu'(.) * gamma
template <int NP>
__global__ __launch_bounds__(256) void gn_bwd_small_kernel(
    const half_t* __restrict__ X, int ldx, const half_t* __restrict__ dY, int lddy, half_t* __restrict__ dX, int lddx,
    const half_t* __restrict__ R, int ldr, int HW, int C, int groups, const float* __restrict__ stats,
    const half_t* __restrict__ gamma, const half_t* __restrict__ beta, int silu) {
  __shared__ float red[8];
  const int g = blockIdx.x, row = blockIdx.y, tid = threadIdx.x;
  const int cpg = C / groups, ppp = cpg >> 3;
  const int npieces = HW * ppp;
  const size_t r0 = (size_t)row * HW;
  const float mean = stats[((size_t)row * groups + g) * 2], rstd = stats[((size_t)row * groups + g) * 2 + 1];
  half8_t xv[NP];
  float d[NP][8];
  int off[NP];
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    const int pi = tid + k * 256;
    const int px = pi / ppp, pc = pi - px * ppp;
    const bool ok = pi < npieces;
    off[k] = ok ? (px << 8) | pc : -1;
    const int c0 = g * cpg + pc * 8;
    xv[k] = ok ? ld_half8(X + (r0 + px) * ldx + c0) : zero_half8();
    const half8_t dv = ok ? ld_half8(dY + (r0 + px) * lddy + c0) : zero_half8();
    const half8_t gv = ok ? ld_half8(gamma + c0) : zero_half8(), bv = ok ? ld_half8(beta + c0) : zero_half8();
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = ((float)xv[k][j] - mean) * rstd;
      float t = (float)dv[j];
      if (silu) t *= silu_grad_f(xh * (float)gv[j] + (float)bv[j]);
      t *= (float)gv[j];
      d[k][j] = t;
      s1 += t; s2 += t * xh;
    }
  }
  const float inv_n = 1.f / ((float)HW * cpg);
  const float m1 = block_sum<256>(s1, red) * inv_n;
  const float m2 = block_sum<256>(s2, red) * inv_n;
#pragma unroll
  for (int k = 0; k < NP; ++k) {
    if (off[k] < 0) continue;
    const int px = off[k] >> 8, pc = off[k] & 255;
    const int c0 = g * cpg + pc * 8;
    const half8_t rv = R ? ld_half8(R + (r0 + px) * ldr + c0) : zero_half8();
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float xh = ((float)xv[k][j] - mean) * rstd;
      o[j] = (half_t)(rstd * (d[k][j] - m1 - xh * m2) + (float)rv[j]);
    }
    st_half8(dX + (r0 + px) * lddx + c0, o);
  }
}

static int gn_fwd_impl(const void* X, const void* Xl, int ldx, void* Y, void* Yl, int ldy, int rows, int HW, int C, int groups,
                       float eps, const void* gamma, const void* beta, int silu, float* stats, float* partial, void* stream) {
  SKG_REQUIRE(X && Y && stats && partial && gamma && beta && rows > 0 && HW > 0 && groups > 0 && groups <= 64);
  SKG_REQUIRE(C % 8 == 0 && C % groups == 0 && (C / groups) % 2 == 0 && (C / groups) >= 4 && C <= GN_MAX_C);
  SKG_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C);
  SKG_REQUIRE(skg_aligned(X, 16) && skg_aligned(Y, 16) && skg_aligned(gamma, 16) && skg_aligned(beta, 16) && skg_aligned(Xl, 16));
  hipStream_t st = (hipStream_t)stream;
  const int cpg = C / groups;
  if (cpg % 8 == 0 && (cpg >> 3) <= 255 && (long)HW * (cpg >> 3) <= 2560) {      // the slice fits 256 threads' registers
    const int np = skg_cdiv(HW * (cpg >> 3), 256);
    const dim3 grid(groups, rows);
#define SKG_GN_SMALL(NP, H)                                                                                                    \
    hipLaunchKernelGGL((gn_small_kernel<NP, H>), grid, dim3(256), 0, st, (const half_t*)X, (const half_t*)Xl, ldx, (half_t*)Y, \
                       (half_t*)Yl, ldy, HW, C, groups, (const half_t*)gamma, (const half_t*)beta, silu, eps, stats)
#define SKG_GN_SMALL_NP(H)                \
    if (np <= 2) SKG_GN_SMALL(2, H);      \
    else if (np <= 3) SKG_GN_SMALL(3, H); \
    else if (np <= 5) SKG_GN_SMALL(5, H); \
    else SKG_GN_SMALL(10, H)
    if (Xl) { SKG_GN_SMALL_NP(true); } else { SKG_GN_SMALL_NP(false); }
#undef SKG_GN_SMALL_NP
#undef SKG_GN_SMALL
    SKG_CHECK_LAUNCH("skg_groupnorm_fwd (small)");
    return SKG_OK;
  }
  const int nch = gn_chunks(HW);
  hipLaunchKernelGGL((gn_partial_kernel<0>), dim3(nch, rows), dim3(256), 0, st, (const half_t*)X, ldx,
                     (const half_t*)nullptr, 0, HW, C, groups, (const float*)nullptr, (const half_t*)nullptr,
                     (const half_t*)nullptr, 0, partial);
#define SKG_GN_FOLD(H)                                                                                                \
  hipLaunchKernelGGL((gn_apply_kernel<true, H>), dim3(gn_apply_chunks(HW, C), rows), dim3(256), 0, st,               \
                     (const half_t*)X, (const half_t*)Xl, ldx, (half_t*)Y, (half_t*)Yl, ldy, HW, C, groups, (const float*)nullptr, \
                     (const half_t*)gamma, (const half_t*)beta, silu, (const float*)partial, nch,                    \
                     1.f / ((float)HW * (C / groups)), eps, stats)
  if (Xl) SKG_GN_FOLD(true); else SKG_GN_FOLD(false);
#undef SKG_GN_FOLD
  SKG_CHECK_LAUNCH("skg_groupnorm_fwd");
  return SKG_OK;
}

extern "C" int skg_groupnorm_wino_fwd(const void* X, int ldx, void* V, int rows, int IH, int IW, int C, int groups, float eps,
                                      const void* gamma, const void* beta, int silu, float* stats, void* stream) {
  SKG_REQUIRE(X && V && stats && gamma && beta && rows > 0 && IH > 0 && IW > 0 && groups > 0 && groups <= 64);
  SKG_REQUIRE(C % 8 == 0 && C % groups == 0 && ldx % 8 == 0 && ldx >= C);
  SKG_REQUIRE(skg_aligned(X, 16) && skg_aligned(V, 16) && skg_aligned(gamma, 16) && skg_aligned(beta, 16));
  const int cpg = C / groups, HW = IH * IW;
  // one workgroup holds the (row, group) slice in registers, then in LDS: the shapes gn_small_kernel takes, on even maps
  if ((IH & 1) || (IW & 1) || cpg % 8 != 0 || (cpg >> 3) > 255 || (long)HW * (cpg >> 3) > 2560 || (size_t)HW * cpg * 2 > 64 * 1024)
    return SKG_E_UNSUPPORTED;
  const int np = skg_cdiv(HW * (cpg >> 3), 256);
  const dim3 grid(groups, rows);
  const size_t lds = (size_t)HW * cpg * 2;
  hipStream_t st = (hipStream_t)stream;
#define SKG_GN_WINO(NP)                                                                                                          \
  hipLaunchKernelGGL((gn_small_wino_kernel<NP>), grid, dim3(256), lds, st, (const half_t*)X, ldx, (half_t*)V, IH, IW, C, groups, \
                     (const half_t*)gamma, (const half_t*)beta, silu, eps, stats)
  if (np <= 2) SKG_GN_WINO(2);
  else if (np <= 3) SKG_GN_WINO(3);
  else if (np <= 5) SKG_GN_WINO(5);
  else SKG_GN_WINO(10);
#undef SKG_GN_WINO
  SKG_CHECK_LAUNCH("skg_groupnorm_wino_fwd");
  return SKG_OK;
}

extern "C" int skg_groupnorm_fwd(const void* X, int ldx, void* Y, int ldy, int rows, int HW, int C, int groups,
                                 float eps, const void* gamma, const void* beta, int silu, float* stats,
                                 float* partial, void* stream) {
  return gn_fwd_impl(X, nullptr, ldx, Y, nullptr, ldy, rows, HW, C, groups, eps, gamma, beta, silu, stats, partial, stream);
}

// accuracy mode: GroupNorm(+SiLU) of the pair X + X_lo (one pitch) in one or two launches, statistics published; Y_lo != NULL:
// the OUTPUT is a pair too (pitch ldy) - the normalised activation in front of conv_out, whose rounding would reach eps 1 : 1
extern "C" int skg_groupnorm_fwd_hilo(const void* X, const void* X_lo, int ldx, void* Y, void* Y_lo, int ldy, int rows, int HW, int C,
                                      int groups, float eps, const void* gamma, const void* beta, int silu, float* stats,
                                      float* partial, void* stream) {
  SKG_REQUIRE(X_lo);
  return gn_fwd_impl(X, X_lo, ldx, Y, Y_lo, ldy, rows, HW, C, groups, eps, gamma, beta, silu, stats, partial, stream);
}

// the statistics pass alone, nch chunks per sample (the fallback behind skg_gemm_f16_gn / skg_conv3x3_f16_gn when the
// producer's tile cannot write the partial sums itself)
void skg_gn_partial_launch(const half_t* X, int ldx, int rows, int HW, int C, int groups, int nch, float* partial,
                           hipStream_t st) {
  hipLaunchKernelGGL((gn_partial_kernel<0>), dim3(nch, rows), dim3(256), 0, st, X, ldx, (const half_t*)nullptr, 0, HW,
                     C, groups, (const float*)nullptr, (const half_t*)nullptr, (const half_t*)nullptr, 0, partial);
}

// GroupNorm forward from partial sums somebody else produced (a producer epilogue: skg_gemm_f16_gn, skg_conv3x3_f16_gn):
// one launch - the apply kernel folds the nch chunk partials per (row, group) itself and publishes (mean, rstd).
// partialB != NULL: GroupNorm of a concatenation [A (CA channels) | B (C - CA channels)] whose halves were written by two
// producers that each left partial sums behind (groupsA / groupsB groups per chunk over their own channels).  The
// concatenation's group width must be a multiple of both source group widths and CA a multiple of it (e.g. 320 + 320 or
// 640 + 640 channels with 32 groups each way: two source groups per output group); otherwise SKG_E_UNSUPPORTED.
static int gn_from_partial_impl(const void* X, const void* Xl, int ldx, void* Y, void* Yl, int ldy, int rows, int HW, int C, int CA,
                                int groups, float eps, const void* gamma, const void* beta, int silu, float* stats,
                                const float* partialA, int groupsA, const float* partialB, int groupsB, int nch,
                                void* stream) {
  SKG_REQUIRE(X && Y && stats && partialA && gamma && beta && rows > 0 && HW > 0 && groups > 0 && groups <= 64);
  SKG_REQUIRE(nch > 0 && nch <= GN_MAX_CHUNKS);
  SKG_REQUIRE(C % 8 == 0 && C % groups == 0 && (C / groups) % 2 == 0 && (C / groups) >= 4 && C <= GN_MAX_C);
  SKG_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C);
  SKG_REQUIRE(skg_aligned(X, 16) && skg_aligned(Y, 16) && skg_aligned(gamma, 16) && skg_aligned(beta, 16) && skg_aligned(Xl, 16));
  const int cpg = C / groups;
  GnSrc src{nullptr, 0, 0, 1, 1};
  if (partialB) {
    SKG_REQUIRE(CA > 0 && CA < C && groupsA > 0 && groupsB > 0);
    const int CB = C - CA;
    if (CA % cpg != 0 || CA % groupsA != 0 || CB % groupsB != 0) return SKG_E_UNSUPPORTED;
    const int cpgA = CA / groupsA, cpgB = CB / groupsB;
    if (cpg % cpgA != 0 || cpg % cpgB != 0) return SKG_E_UNSUPPORTED;
    src = GnSrc{partialB, CA / cpg, 0, cpg / cpgA, cpg / cpgB};
  }
#define SKG_GN_FOLD(H)                                                                                                          \
  hipLaunchKernelGGL((gn_apply_kernel<true, H>), dim3(gn_apply_chunks(HW, C), rows), dim3(256), 0, (hipStream_t)stream,        \
                     (const half_t*)X, (const half_t*)Xl, ldx, (half_t*)Y, (half_t*)Yl, ldy, HW, C, groups, (const float*)nullptr, \
                     (const half_t*)gamma, (const half_t*)beta, silu, partialA, nch, 1.f / ((float)HW * cpg), eps, stats, src)
  if (Xl) SKG_GN_FOLD(true); else SKG_GN_FOLD(false);
#undef SKG_GN_FOLD
  SKG_CHECK_LAUNCH("skg_groupnorm_from_partial");
  return SKG_OK;
}

extern "C" int skg_groupnorm_from_partial(const void* X, int ldx, void* Y, int ldy, int rows, int HW, int C, int groups,
                                          float eps, const void* gamma, const void* beta, int silu, float* stats,
                                          const float* partial, int nch, void* stream) {
  return gn_from_partial_impl(X, nullptr, ldx, Y, nullptr, ldy, rows, HW, C, 0, groups, eps, gamma, beta, silu, stats, partial, groups,
                              nullptr, 0, nch, stream);
}

extern "C" int skg_groupnorm_from_partial2(const void* X, int ldx, void* Y, int ldy, int rows, int HW, int C, int CA,
                                           int groups, float eps, const void* gamma, const void* beta, int silu,
                                           float* stats, const float* partialA, int groupsA, const float* partialB,
                                           int groupsB, int nch, void* stream) {
  SKG_REQUIRE(partialB);
  return gn_from_partial_impl(X, nullptr, ldx, Y, nullptr, ldy, rows, HW, C, CA, groups, eps, gamma, beta, silu, stats, partialA, groupsA,
                              partialB, groupsB, nch, stream);
}

// accuracy mode: the same for a pair X + X_lo (partialB == NULL: one producer; else the concatenation form)
extern "C" int skg_groupnorm_from_partial_hilo(const void* X, const void* X_lo, int ldx, void* Y, void* Y_lo, int ldy, int rows, int HW,
                                               int C, int CA, int groups, float eps, const void* gamma, const void* beta,
                                               int silu, float* stats, const float* partialA, int groupsA,
                                               const float* partialB, int groupsB, int nch, void* stream) {
  SKG_REQUIRE(X_lo && (partialB || groupsA == groups));
  return gn_from_partial_impl(X, X_lo, ldx, Y, Y_lo, ldy, rows, HW, C, CA, groups, eps, gamma, beta, silu, stats, partialA, groupsA,
                              partialB, groupsB, nch, stream);
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
  const int cpg = C / groups;
  if (cpg % 8 == 0 && (cpg >> 3) <= 255 && (long)HW * (cpg >> 3) <= 2560) {      // one launch: see gn_small_kernel
    const int np = skg_cdiv(HW * (cpg >> 3), 256);
    const dim3 grid(groups, rows);
#define SKG_GN_BWD_SMALL(NP)                                                                                         \
    hipLaunchKernelGGL((gn_bwd_small_kernel<NP>), grid, dim3(256), 0, st, (const half_t*)X, ldx, (const half_t*)dY,   \
                       lddy, (half_t*)dX, lddx, (const half_t*)residual, ldr, HW, C, groups, stats,                   \
                       (const half_t*)gamma, (const half_t*)beta, silu)
    if (np <= 2) SKG_GN_BWD_SMALL(2);
    else if (np <= 3) SKG_GN_BWD_SMALL(3);
    else if (np <= 5) SKG_GN_BWD_SMALL(5);
    else SKG_GN_BWD_SMALL(10);
#undef SKG_GN_BWD_SMALL
    SKG_CHECK_LAUNCH("skg_groupnorm_bwd (small)");
    return SKG_OK;
  }
  const int nch = gn_chunks(HW);
  hipLaunchKernelGGL((gn_partial_kernel<1>), dim3(nch, rows), dim3(256), 0, st, (const half_t*)X, ldx,
                     (const half_t*)dY, lddy, HW, C, groups, stats, (const half_t*)gamma, (const half_t*)beta,
                     silu, partial);
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(gn_apply_chunks(HW, C), rows), dim3(256), 0, st, (const half_t*)X,
                     ldx, (const half_t*)dY, lddy, (half_t*)dX, lddx, (const half_t*)residual, ldr, HW, C, groups,
                     stats, (const float*)partial, nch, 1.f / ((float)HW * (C / groups)), (const half_t*)gamma, (const half_t*)beta, silu);
  SKG_CHECK_LAUNCH("skg_groupnorm_bwd");
  return SKG_OK;
}

static int ln_fwd_impl(const void* X, const void* Xl, int ldx, void* Y, int ldy, int M, int C, const void* gamma,
                       const void* beta, float eps, float* stats, void* stream) {
  SKG_REQUIRE(X && Y && gamma && beta && M > 0 && C % 8 == 0 && C <= LN_MAXP * 64 * 8);
  SKG_REQUIRE(ldx % 8 == 0 && ldy % 8 == 0 && skg_aligned(X, 16) && skg_aligned(Y, 16) && skg_aligned(Xl, 16) &&
              skg_aligned(gamma, 16) && skg_aligned(beta, 16));
  hipStream_t st = (hipStream_t)stream;
  const half_t *x = (const half_t*)X, *xl = (const half_t*)Xl, *g = (const half_t*)gamma, *b = (const half_t*)beta;
  // RW rows per wave so that four independent 16-byte loads are in flight per lane (pair input: two tensors per row)
#define SKG_LN(NQ, RW, H) \
  hipLaunchKernelGGL((ln_fwd_kernel<NQ, RW, H>), dim3(skg_cdiv(M, 4 * RW)), dim3(256), 0, st, x, xl, ldx, (half_t*)Y, ldy, M, C, g, b, eps, stats)
  if (Xl) {
    if (C <= 512) SKG_LN(1, 2, true);
    else if (C <= 1024) SKG_LN(2, 1, true);
    else SKG_LN(LN_MAXP, 1, true);
  } else {
    if (C <= 512) SKG_LN(1, 4, false);
    else if (C <= 1024) SKG_LN(2, 2, false);
    else SKG_LN(LN_MAXP, 1, false);
  }
#undef SKG_LN
  SKG_CHECK_LAUNCH("skg_layernorm_fwd");
  return SKG_OK;
}

extern "C" int skg_layernorm_fwd(const void* X, int ldx, void* Y, int ldy, int M, int C, const void* gamma,
                                 const void* beta, float eps, float* stats, void* stream) {
  return ln_fwd_impl(X, nullptr, ldx, Y, ldy, M, C, gamma, beta, eps, stats, stream);
}

extern "C" int skg_layernorm_fwd_hilo(const void* X, const void* X_lo, int ldx, void* Y, int ldy, int M, int C,
                                      const void* gamma, const void* beta, float eps, float* stats, void* stream) {
  SKG_REQUIRE(X_lo);
  return ln_fwd_impl(X, X_lo, ldx, Y, ldy, M, C, gamma, beta, eps, stats, stream);
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

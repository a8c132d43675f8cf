// LGP (Latent Gradient/Edge Predictor) kernels: re-associated layer 0 (bilinear gather of per-tap
// partial products + noise-level / sinusoid channels), its adjoint, per-sample train-mode
// BatchNorm1d (stats / apply / backward through the preceding ReLU) and the MSE seed gradient.
//
// Row layout of every LGP activation: row = (j * S + s) * hw + pixel with j = CFG half (0 = uncond,
// 1 = cond), s = sample, pixel = y * h + x.  One BatchNorm "batch" is one sample's two CFG rows
// (2*hw LGP rows): the reference only runs B = 1, so statistics are per sample.
#include "common.h"

namespace {

constexpr int BN_CHUNKS = 64;
constexpr int MAX_TAPS = 12;
constexpr int NEXTRA = 40;   // 4 noise-level channels + 9 * 4 sinusoid channels

struct TapArgs {
  const float* P[MAX_TAPS];
  int s[MAX_TAPS];
  int n;
};

__device__ __forceinline__ void bil_coord(int d, int s, int h, int& i0, int& i1, float& w1) {
  const float scale = (float)s / (float)h;
  float src = ((float)d + 0.5f) * scale - 0.5f;
  if (src < 0.f) src = 0.f;
  i0 = (int)src;
  if (i0 > s - 1) i0 = s - 1;
  i1 = i0 + (i0 < s - 1 ? 1 : 0);
  w1 = src - (float)i0;
}

// one 128-thread half-block per output pixel, 4 channels per thread (H0 == 512)
__global__ __launch_bounds__(256) void lgp_gather_kernel(const TapArgs taps, const half_t* __restrict__ Wx, int ldw,
                                                         const half_t* __restrict__ bias0,
                                                         const float* __restrict__ noise, float sigma, int S,
                                                         half_t* __restrict__ Z, int rows, int h, int H0) {
  __shared__ float e_s[2][NEXTRA];
  const int hw = h * h;
  const int half_id = threadIdx.x >> 7;
  const int tl = threadIdx.x & 127;
  const size_t pix = (size_t)blockIdx.x * 2 + half_id;   // over rows*hw
  const bool ok = pix < (size_t)rows * hw;
  const int row = ok ? (int)(pix / hw) : 0;
  const int pp = ok ? (int)(pix - (size_t)row * hw) : 0;
  const int y = pp / h, x = pp - y * h;
  const int smp = row % S;
  if (tl < NEXTRA) {
    const int c = tl < 4 ? tl : (tl - 4) & 3;
    const float nl = sigma * noise[((size_t)smp * 4 + c) * hw + pp];
    float v = nl;
    if (tl >= 4) {
      const int l = (tl - 4) >> 2;
      v = sinf((6.283185307179586f * nl) * exp2f(-(float)l));
    }
    e_s[half_id][tl] = (float)(half_t)v;    // the reference casts the concatenated input to fp16
  }
  __syncthreads();
  if (!ok) return;
  for (int c0 = tl * 4; c0 < H0; c0 += 512) {
    float4_t acc = {0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < taps.n; ++i) {
      const int s = taps.s[i];
      const float* P = taps.P[i] + (size_t)row * s * s * H0 + c0;
      if (s == h) {
        acc += *reinterpret_cast<const float4_t*>(P + (size_t)pp * H0);
      } else {
        int y0, y1, x0, x1;
        float wy, wx;
        bil_coord(y, s, h, y0, y1, wy);
        bil_coord(x, s, h, x0, x1, wx);
        const float4_t v00 = *reinterpret_cast<const float4_t*>(P + (size_t)(y0 * s + x0) * H0);
        const float4_t v01 = *reinterpret_cast<const float4_t*>(P + (size_t)(y0 * s + x1) * H0);
        const float4_t v10 = *reinterpret_cast<const float4_t*>(P + (size_t)(y1 * s + x0) * H0);
        const float4_t v11 = *reinterpret_cast<const float4_t*>(P + (size_t)(y1 * s + x1) * H0);
        acc += (1.f - wy) * ((1.f - wx) * v00 + wx * v01) + wy * ((1.f - wx) * v10 + wx * v11);
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const half_t* w = Wx + (size_t)(c0 + j) * ldw;
      float d = 0.f;
#pragma unroll
      for (int k = 0; k < NEXTRA; k += 8) {
        const half8_t wv = ld_half8(w + k);
#pragma unroll
        for (int q = 0; q < 8; ++q) d += (float)wv[q] * e_s[half_id][k + q];
      }
      acc[j] += d + (float)bias0[c0 + j];
    }
    half4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = (half_t)fmaxf((float)(half_t)acc[j], 0.f);
    st_half4(Z + pix * H0 + c0, o);
  }
}

// Tiled variant (h % 8 == 0, H0 % 128 == 0): one workgroup = an 8x8 output tile x 128 channels of one row.  For every
// tap the source window of the tile (<= 8x8 native pixels; 3x3 for an 8x-upsampled tap) is staged ONCE in LDS and
// every thread interpolates from there: the per-pixel kernel above re-read four 2 KB corner rows per tap and output
// pixel from L2 (73 KB per output pixel, ~8 TB/s of L2 traffic for 589 us at 64x64); here the taps cost ~5 KB per
// output pixel.  thread = (output pixel = tid / 4, 32-channel group = tid % 4).
constexpr int GT = 8, GC = 128, GPITCH = GC + 4;
__global__ __launch_bounds__(256) void lgp_gather_tiled_kernel(const TapArgs taps, const half_t* __restrict__ Wx, int ldw,
                                                               const half_t* __restrict__ bias0,
                                                               const float* __restrict__ noise, float sigma, int S,
                                                               half_t* __restrict__ Z, int rows, int h, int H0) {
  __shared__ __attribute__((aligned(16))) float patch[GT * GT * GPITCH];
  __shared__ float e_s[GT * GT][NEXTRA];
  const int hw = h * h;
  const int tiles = h / GT;
  const int ty = blockIdx.x / tiles, tx = blockIdx.x - ty * tiles;
  const int row = blockIdx.y, cbase = blockIdx.z * GC;
  const int tid = threadIdx.x;
  const int pix = tid >> 2, cg = tid & 3;
  const int py = pix >> 3, px = pix & 7;
  const int oy = ty * GT + py, ox = tx * GT + px;
  const int smp = row % S;
  for (int i = tid; Wx != nullptr && i < GT * GT * NEXTRA; i += 256) {
    const int p = i / NEXTRA, tl = i - p * NEXTRA;
    const int pp = (ty * GT + (p >> 3)) * h + tx * GT + (p & 7);
    const int c = tl < 4 ? tl : (tl - 4) & 3;
    const float nl = sigma * noise[((size_t)smp * 4 + c) * hw + pp];
    float v = nl;
    if (tl >= 4) v = sinf((6.283185307179586f * nl) * exp2f(-(float)((tl - 4) >> 2)));
    e_s[p][tl] = (float)(half_t)v;          // the reference casts the concatenated input to fp16
  }
  float4_t acc[8];
#pragma unroll
  for (int q = 0; q < 8; ++q) acc[q] = float4_t{0.f, 0.f, 0.f, 0.f};
  for (int i = 0; i < taps.n; ++i) {
    const int s = taps.s[i];
    int sy0, sx0, nrow, ncol;
    if (s == h) {
      sy0 = ty * GT; sx0 = tx * GT; nrow = ncol = GT;
    } else {
      int a0, a1, b0, b1;
      float w;
      bil_coord(ty * GT, s, h, a0, a1, w);
      bil_coord(ty * GT + GT - 1, s, h, b0, b1, w);
      sy0 = a0; nrow = b1 - a0 + 1;
      bil_coord(tx * GT, s, h, a0, a1, w);
      bil_coord(tx * GT + GT - 1, s, h, b0, b1, w);
      sx0 = a0; ncol = b1 - a0 + 1;
    }
    __syncthreads();                                   // the previous tap's window has been consumed
    const float* P = taps.P[i] + (size_t)row * s * s * H0 + cbase;
    for (int idx = tid; idx < nrow * ncol * (GC / 4); idx += 256) {
      const int sp = idx / (GC / 4), q = idx - sp * (GC / 4);
      const int sy = sp / ncol, sx = sp - sy * ncol;
      *reinterpret_cast<float4_t*>(&patch[sp * GPITCH + q * 4]) =
          *reinterpret_cast<const float4_t*>(P + (size_t)((sy0 + sy) * s + sx0 + sx) * H0 + q * 4);
    }
    __syncthreads();
    const float* pc = patch + cg * 32;
    if (s == h) {
      const float* a = pc + (py * GT + px) * GPITCH;
#pragma unroll
      for (int q = 0; q < 8; ++q) acc[q] += *reinterpret_cast<const float4_t*>(a + q * 4);
    } else {
      int y0, y1, x0, x1;
      float wy, wx;
      bil_coord(oy, s, h, y0, y1, wy);
      bil_coord(ox, s, h, x0, x1, wx);
      const float* p00 = pc + ((y0 - sy0) * ncol + (x0 - sx0)) * GPITCH;
      const float* p01 = pc + ((y0 - sy0) * ncol + (x1 - sx0)) * GPITCH;
      const float* p10 = pc + ((y1 - sy0) * ncol + (x0 - sx0)) * GPITCH;
      const float* p11 = pc + ((y1 - sy0) * ncol + (x1 - sx0)) * GPITCH;
#pragma unroll
      for (int q = 0; q < 8; ++q) {
        const float4_t v00 = *reinterpret_cast<const float4_t*>(p00 + q * 4);
        const float4_t v01 = *reinterpret_cast<const float4_t*>(p01 + q * 4);
        const float4_t v10 = *reinterpret_cast<const float4_t*>(p10 + q * 4);
        const float4_t v11 = *reinterpret_cast<const float4_t*>(p11 + q * 4);
        acc[q] += (1.f - wy) * ((1.f - wx) * v00 + wx * v01) + wy * ((1.f - wx) * v10 + wx * v11);
      }
    }
  }
  const size_t orow = (size_t)row * hw + (size_t)oy * h + ox;
  half_t* zo = Z + orow * H0 + cbase + cg * 32;
#pragma unroll
  for (int q = 0; q < 8; ++q) {
    half4_t o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = cbase + cg * 32 + q * 4 + j;
      float d = 0.f;
      if (Wx != nullptr) {       // (callers normally pass the 40 extra channels as one more s == h tap instead)
        const half_t* w = Wx + (size_t)c * ldw;
#pragma unroll
        for (int k = 0; k < NEXTRA; k += 8) {
          const half8_t wv = ld_half8(w + k);
#pragma unroll
          for (int u = 0; u < 8; ++u) d += (float)wv[u] * e_s[pix][k + u];
        }
      }
      const float v = acc[q][j] + d + (float)bias0[c];
      o[j] = (half_t)fmaxf((float)(half_t)v, 0.f);
    }
    st_half4(zo + q * 4, o);
  }
}

// adjoint of the bilinear resize for one tap: one block per native pixel; thread = (8-channel piece, window lane):
// 16-byte loads, the window ROWS are dealt round-robin to the 256 / (H0/8) window lanes and folded in LDS.  The
// bilinear weight is separable: the row / column factors of the (<= 3f)^2 window are computed once per block into LDS,
// so the inner loop is a multiply, a test and a load + 8 FMAs (the first version redid both coordinate computations and
// an integer division for every window pixel: ~40 VALU per 16-byte load, and two thirds of the window has zero weight).
__global__ __launch_bounds__(256) void lgp_scatter_kernel(const half_t* __restrict__ dZ, int lddz,
                                                          half_t* __restrict__ dP, int rows, int h, int s, int H0) {
  __shared__ float red[256][9];
  __shared__ float cys[256], cxs[256];
  const int ss = s * s;
  const int row = blockIdx.x / ss;
  const int np = blockIdx.x - row * ss;
  const int py = np / s, px = np - py * s;
  const int f = h / s;
  const int ylo = max(0, f * py - f), yhi = min(h - 1, f * py + 2 * f - 1);
  const int xlo = max(0, f * px - f), xhi = min(h - 1, f * px + 2 * f - 1);
  const int wy_n = yhi - ylo + 1, wx_n = xhi - xlo + 1;
  const bool tab = wy_n <= 256 && wx_n <= 256;            // (always, for the resolutions of the pipeline)
  auto coef = [&](int v, int p) {
    int v0, v1;
    float w;
    bil_coord(v, s, h, v0, v1, w);
    return (v0 == p ? 1.f - w : 0.f) + (v1 == p ? w : 0.f);
  };
  if (tab) {
    for (int i = threadIdx.x; i < wy_n; i += 256) cys[i] = coef(ylo + i, py);
    for (int i = threadIdx.x; i < wx_n; i += 256) cxs[i] = coef(xlo + i, px);
    __syncthreads();
  }
  const int C8 = H0 >> 3;
  for (int pb = 0; pb < C8; pb += 256) {
    const int npc = min(256, C8 - pb);
    const int WL = 256 / npc;
    const int piece = threadIdx.x % npc, wl = threadIdx.x / npc;
    const int c0 = (pb + piece) * 8;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (wl < WL) {
      for (int yy = wl; yy < wy_n; yy += WL) {
        const float cy = tab ? cys[yy] : coef(ylo + yy, py);
        if (cy == 0.f) continue;
        const half_t* src = dZ + ((size_t)row * h * h + (size_t)(ylo + yy) * h + xlo) * lddz + c0;
        for (int xx = 0; xx < wx_n; ++xx) {
          const float w = cy * (tab ? cxs[xx] : coef(xlo + xx, px));
          if (w == 0.f) continue;
          const half8_t v = ld_half8(src + (size_t)xx * lddz);
#pragma unroll
          for (int j = 0; j < 8; ++j) a[j] += w * (float)v[j];
        }
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = a[j];
    __syncthreads();
    if (threadIdx.x < npc) {
      half8_t o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t = 0.f;
        for (int l = 0; l < WL; ++l) t += red[l * npc + threadIdx.x][j];
        o[j] = (half_t)t;
      }
      st_half8(dP + ((size_t)row * ss + np) * H0 + c0, o);
    }
  }
}

// ---- BatchNorm1d, statistics over one sample's rows -------------------------------------------
// KIND 0: (sum x, sum x^2);  KIND 1: (sum dy, sum dy*xhat)
template <int KIND>
__global__ __launch_bounds__(256) void bn_partial_kernel(const half_t* __restrict__ X, int ldx,
                                                         const half_t* __restrict__ dY, int lddy, int S, int segs,
                                                         int seg_rows, int C, const float* __restrict__ stats,
                                                         float* __restrict__ partial) {
  // thread = (8-channel piece, row lane): 16-byte loads, two rows in flight per iteration, fixed-order LDS fold
  __shared__ float red[256][17];
  const int g = blockIdx.y, chunk = blockIdx.x;
  const int n = segs * seg_rows;
  const int per = (n + BN_CHUNKS - 1) / BN_CHUNKS;
  const int i0 = chunk * per, i1 = min(n, i0 + per);
  const int C8 = C >> 3;
  for (int pb = 0; pb < C8; pb += 256) {               // C <= 2048: one pass
    const int npc = min(256, C8 - pb);                   // pieces in this pass
    const int RL = 256 / npc;                            // row lanes
    const int piece = threadIdx.x % npc, rl = threadIdx.x / npc;
    const int c0 = (pb + piece) * 8;
    float s1[8], s2[8], mu[8], rs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { s1[j] = 0.f; s2[j] = 0.f; mu[j] = 0.f; rs[j] = 0.f; }
    if (rl < RL) {
      if (KIND == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          mu[j] = stats[((size_t)g * C + c0 + j) * 2];
          rs[j] = stats[((size_t)g * C + c0 + j) * 2 + 1];
        }
      }
      auto rowof = [&](int i) {
        const int jseg = i / seg_rows, ii = i - jseg * seg_rows;
        return ((size_t)jseg * S + g) * seg_rows + ii;
      };
      auto accum = [&](const half8_t& xv, const half8_t& dv) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          if (KIND == 0) {
            const float a = (float)xv[j];
            s1[j] += a; s2[j] += a * a;
          } else {
            const float d = (float)dv[j];
            s1[j] += d; s2[j] += d * ((float)xv[j] - mu[j]) * rs[j];
          }
        }
      };
      int i = i0 + rl;
      for (; i + RL < i1; i += 2 * RL) {
        const size_t ra = rowof(i), rb = rowof(i + RL);
        const half8_t xa = ld_half8(X + ra * ldx + c0), xb = ld_half8(X + rb * ldx + c0);
        const half8_t da = KIND == 1 ? ld_half8(dY + ra * lddy + c0) : zero_half8();
        const half8_t db = KIND == 1 ? ld_half8(dY + rb * lddy + c0) : zero_half8();
        accum(xa, da);
        accum(xb, db);
      }
      for (; i < i1; i += RL) {
        const size_t r = rowof(i);
        accum(ld_half8(X + r * ldx + c0), KIND == 1 ? ld_half8(dY + r * lddy + c0) : zero_half8());
      }
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 8; ++j) { red[threadIdx.x][j] = s1[j]; red[threadIdx.x][8 + j] = s2[j]; }
    __syncthreads();
    if (threadIdx.x < npc) {
      float* o = partial + (((size_t)g * BN_CHUNKS + chunk) * C + c0) * 2;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float t1 = 0.f, t2 = 0.f;
        for (int l = 0; l < RL; ++l) { t1 += red[l * npc + threadIdx.x][j]; t2 += red[l * npc + threadIdx.x][8 + j]; }
        o[2 * j] = t1; o[2 * j + 1] = t2;
      }
    }
  }
}

// fold chunks, one (channel, sample) per thread; KIND 0 -> (mean, rstd), KIND 1 -> (m1, m2).  var_out (KIND 0, optional)
// receives the biased variance for the running-statistics update.
template <int KIND>
__global__ void bn_finalize_kernel(const float* __restrict__ partial, int S, int C, int n, float eps,
                                   float* __restrict__ out, float* __restrict__ var_out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  const int g = blockIdx.y;
  if (c >= C) return;
  float s1 = 0.f, s2 = 0.f;
  for (int k = 0; k < BN_CHUNKS; ++k) {
    const float* q = partial + (((size_t)g * BN_CHUNKS + k) * C + c) * 2;
    s1 += q[0]; s2 += q[1];
  }
  if (KIND == 0) {
    const float mean = s1 / n;
    const float var = fmaxf(s2 / n - mean * mean, 0.f);
    out[((size_t)g * C + c) * 2] = mean;
    out[((size_t)g * C + c) * 2 + 1] = rsqrtf(var + eps);
    if (var_out) var_out[(size_t)g * C + c] = var;
  } else {
    out[((size_t)g * C + c) * 2] = s1 / n;
    out[((size_t)g * C + c) * 2 + 1] = s2 / n;
  }
}

// running statistics: the S samples of a call are S successive BatchNorm batches (momentum 0.1, unbiased variance),
// applied in sample order exactly like S separate module calls
__global__ void bn_running_kernel(const float* __restrict__ stats, const float* __restrict__ var, int S, int C, int n,
                                  float* __restrict__ running_mean, float* __restrict__ running_var) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float rm = running_mean[c], rv = running_var[c];
  for (int g = 0; g < S; ++g) {
    rm = 0.9f * rm + 0.1f * stats[((size_t)g * C + c) * 2];
    rv = 0.9f * rv + 0.1f * var[(size_t)g * C + c] * ((float)n / (float)(n - 1));
  }
  running_mean[c] = rm;
  running_var[c] = rv;
}

__global__ void bn_from_running_kernel(const float* __restrict__ rm, const float* __restrict__ rv, int S, int C,
                                       float eps, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= S * C) return;
  const int c = i % C;
  out[(size_t)i * 2] = rm[c];
  out[(size_t)i * 2 + 1] = rsqrtf(rv[c] + eps);
}

// MODE 0: y = (x-mean)*rstd*gamma+beta.   MODE 1: dx = [x>0] * gamma*rstd*(dy - m1 - xhat*m2)
// Workgroup = (row chunk, sample), thread = fixed 8-channel piece x strided rows (the decomposition of bn_partial_kernel):
// the per-channel coefficients are loop-invariant registers -  y = x*a + sh,  dx = [x>0] * (dy*a - b - x*c)  - and four
// rows are in flight per thread.  (The first version re-read 128 bytes of statistics per 16 bytes of x through the
// vector-memory path, which capped it at 1.3 TB/s.)
// row chunks per sample: >= 8 rows per thread and chunk, <= 512 chunks
inline int bn_apply_chunks(int n, int C) {
  const int C8 = C >> 3, RL = C8 >= 256 ? 1 : 256 / C8;
  int c = n / (8 * RL);
  return c < 1 ? 1 : (c > 512 ? 512 : c);
}
template <int MODE>
__global__ __launch_bounds__(256) void bn_apply_kernel(const half_t* __restrict__ X, int ldx,
                                                       const half_t* __restrict__ dY, int lddy,
                                                       half_t* __restrict__ Y, int ldy, int S, int segs, int seg_rows,
                                                       int C, const float* __restrict__ stats,
                                                       const float* __restrict__ sums,
                                                       const half_t* __restrict__ gamma,
                                                       const half_t* __restrict__ beta, int train) {
  const int g = blockIdx.y, chunk = blockIdx.x;
  const int n = segs * seg_rows;
  const int per = (n + (int)gridDim.x - 1) / (int)gridDim.x;
  const int i0 = chunk * per, i1 = min(n, i0 + per);
  const int C8 = C >> 3;
  auto rowof = [&](int i) {
    const int jseg = i / seg_rows, ii = i - jseg * seg_rows;
    return ((size_t)jseg * S + g) * seg_rows + ii;
  };
  for (int pb = 0; pb < C8; pb += 256) {
    const int npc = min(256, C8 - pb);
    const int RL = 256 / npc;
    const int piece = threadIdx.x % npc, rl = threadIdx.x / npc;
    if (rl >= RL) continue;
    const int c0 = (pb + piece) * 8;
    float ca[8], cb[8], cc[8];
    {
      const half8_t gv = ld_half8(gamma + c0);
      const half8_t bv = MODE == 0 ? ld_half8(beta + c0) : zero_half8();
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float mean = stats[((size_t)g * C + c0 + j) * 2], rstd = stats[((size_t)g * C + c0 + j) * 2 + 1];
        const float a = (float)gv[j] * rstd;
        ca[j] = a;
        if (MODE == 0) {
          cb[j] = (float)bv[j] - mean * a;
          cc[j] = 0.f;
        } else {
          const float m1 = train ? sums[((size_t)g * C + c0 + j) * 2] : 0.f;
          const float m2 = train ? sums[((size_t)g * C + c0 + j) * 2 + 1] : 0.f;
          cc[j] = a * rstd * m2;                       // dx = [x>0] * (a*dy - a*m1 - a*rstd*m2*(x - mean))
          cb[j] = a * m1 - cc[j] * mean;
        }
      }
    }
    auto apply = [&](const half8_t& xv, const half8_t& dv) {
      half8_t o;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float x = (float)xv[j];
        if (MODE == 0) o[j] = (half_t)(x * ca[j] + cb[j]);
        else o[j] = x > 0.f ? (half_t)((float)dv[j] * ca[j] - cb[j] - x * cc[j]) : (half_t)0.f;
      }
      return o;
    };
    int i = i0 + rl;
    for (; i + 3 * RL < i1; i += 4 * RL) {
      size_t r[4];
      half8_t xv[4], dv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        r[u] = rowof(i + u * RL);
        xv[u] = ld_half8(X + r[u] * ldx + c0);
        dv[u] = MODE == 1 ? ld_half8(dY + r[u] * lddy + c0) : zero_half8();
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) st_half8(Y + r[u] * ldy + c0, apply(xv[u], dv[u]));
    }
    for (; i < i1; i += RL) {
      const size_t r = rowof(i);
      st_half8(Y + r * ldy + c0, apply(ld_half8(X + r * ldx + c0), MODE == 1 ? ld_half8(dY + r * lddy + c0) : zero_half8()));
    }
  }
}

// one block per sample: loss, and the seed gradient for the cond row.  dOut has been zero-filled by the launcher
// (uncond rows and the padding columns stay zero); this kernel only writes the 4 real channels of the cond rows,
// one pixel (8 bytes) per thread and iteration.
__global__ __launch_bounds__(256) void mse_seed_kernel(const half_t* __restrict__ out, int ldo,
                                                       const float* __restrict__ target, half_t* __restrict__ dOut,
                                                       int ldd, float* __restrict__ loss, int S, int h,
                                                       float loss_scale) {
  __shared__ float red[8];
  const int s = blockIdx.x;
  const int hw = h * h;
  const float k = loss_scale * 2.f / (4.f * hw);
  float acc = 0.f;
  for (int p = threadIdx.x; p < hw; p += 256) {
    const size_t crow = ((size_t)S + s) * hw + p;
    const half4_t o = ld_half4(out + crow * ldo);
    half4_t gq;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float d = (float)o[c] - target[((size_t)s * 4 + c) * hw + p];
      acc += d * d;
      gq[c] = (half_t)(k * d);
    }
    st_half4(dOut + crow * ldd, gq);
  }
  acc = block_sum<256>(acc, red);
  if (threadIdx.x == 0 && loss) loss[s] = acc / (4.f * hw);
}

// ---- training-only kernels (LGP weight gradients + AdamW: SURVEY 8f row 4, trainer.py:208-252) --------------------
// column sums of a [M][C] fp16 matrix (bias gradients): block (x = 8-channel piece, y = row chunk) -> partial[chunk][C]
constexpr int CS_CHUNKS = 32;
__global__ __launch_bounds__(256) void colsum_partial_kernel(const half_t* __restrict__ X, int ldx, int M, int C,
                                                             float* __restrict__ partial) {
  __shared__ float red[256][8];
  const int c0 = blockIdx.x * 8, chunk = blockIdx.y;
  const int per = (M + CS_CHUNKS - 1) / CS_CHUNKS;
  const int r0 = chunk * per, r1 = min(M, r0 + per);
  float a[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int r = r0 + threadIdx.x; r < r1; r += 256) {
    const half8_t v = ld_half8(X + (size_t)r * ldx + c0);
#pragma unroll
    for (int j = 0; j < 8; ++j) a[j] += (float)v[j];
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) red[threadIdx.x][j] = a[j];
  __syncthreads();
  if (threadIdx.x < 8) {          // fixed-order fold: deterministic
    float t = 0.f;
    for (int i = 0; i < 256; ++i) t += red[i][threadIdx.x];
    partial[(size_t)chunk * C + c0 + threadIdx.x] = t;
  }
}
__global__ void colsum_final_kernel(const float* __restrict__ partial, int C, float scale, float* __restrict__ out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float t = 0.f;
  for (int k = 0; k < CS_CHUNKS; ++k) t += partial[(size_t)k * C + c];
  out[c] = t * scale;
}

// (d gamma, d beta) = (sum dy*xhat, sum dy) * scale from the folded chunk partials of bn_partial_kernel<1>
__global__ void bn_param_grads_kernel(const float* __restrict__ partial, int C, float scale,
                                      float* __restrict__ dgamma, float* __restrict__ dbeta) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s1 = 0.f, s2 = 0.f;
  for (int k = 0; k < BN_CHUNKS; ++k) {
    const float* q = partial + ((size_t)k * C + c) * 2;
    s1 += q[0]; s2 += q[1];
  }
  dbeta[c] = s1 * scale;
  dgamma[c] = s2 * scale;
}

// the 40 extra input channels of layer 0 (noise level + 36 sinusoids), fp16-rounded like the reference's cast:
// E [rows*hw][ld] with columns >= 40 zero.  Same arithmetic as lgp_gather_kernel.
__global__ __launch_bounds__(256) void lgp_extra_kernel(const float* __restrict__ noise, float sigma, int S, int rows,
                                                        int h, half_t* __restrict__ E, int ld) {
  const int hw = h * h;
  const size_t total = (size_t)rows * hw * ld;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t pix = i / ld;
    const int tl = (int)(i - pix * ld);
    float v = 0.f;
    if (tl < NEXTRA) {
      const int row = (int)(pix / hw), pp = (int)(pix - (size_t)row * hw);
      const int smp = row % S;
      const int c = tl < 4 ? tl : (tl - 4) & 3;
      const float nl = sigma * noise[((size_t)smp * 4 + c) * hw + pp];
      v = nl;
      if (tl >= 4) v = sinf((6.283185307179586f * nl) * exp2f(-(float)((tl - 4) >> 2)));
    }
    E[i] = (half_t)v;
  }
}

// training loss: one scalar MSE over all samples; dOut = loss_scale * 2 (out - target) / (S*4*hw) on every row
__global__ __launch_bounds__(256) void mse_train_kernel(const half_t* __restrict__ out, int ldo,
                                                        const float* __restrict__ target, half_t* __restrict__ dOut,
                                                        int ldd, float* __restrict__ loss_part, int S, int h,
                                                        float loss_scale) {
  __shared__ float red[8];
  const int s = blockIdx.x;
  const int hw = h * h;
  const float n = 4.f * hw * S;
  const float k = loss_scale * 2.f / n;
  float acc = 0.f;
  for (int p = threadIdx.x; p < hw; p += 256) {
    const size_t row = (size_t)s * hw + p;
    const half4_t o = ld_half4(out + row * ldo);
    half4_t gq;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const float d = (float)o[c] - target[((size_t)s * 4 + c) * hw + p];
      acc += d * d;
      gq[c] = (half_t)(k * d);
    }
    st_half4(dOut + row * ldd, gq);
  }
  acc = block_sum<256>(acc, red);
  if (threadIdx.x == 0) loss_part[s] = acc / n;       // the caller sums the S partials (fixed order)
}

// AdamW (decoupled weight decay), fp32 master weights + fp16 working copy; g is the loss-scaled gradient
__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v,
                                                    half_t* __restrict__ p16, size_t n, float lr, float b1, float b2,
                                                    float eps, float wd, float bc1, float bc2, float inv_scale) {
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
    const float gi = g[i] * inv_scale;
    const float mi = b1 * m[i] + (1.f - b1) * gi;
    const float vi = b2 * v[i] + (1.f - b2) * gi * gi;
    m[i] = mi;
    v[i] = vi;
    float w = p[i] * (1.f - lr * wd);
    w -= lr * (mi / bc1) / (sqrtf(vi / bc2) + eps);
    p[i] = w;
    if (p16) p16[i] = (half_t)w;
  }
}

inline int ew_grid(size_t total_items) {
  size_t b = (total_items + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int skg_lgp_layer0_gather(const SkgLgpTap* taps, int ntaps, const void* Wextra, int ldw,
                                     const void* bias0, const float* noise, float sigma, int samples, void* Z,
                                     int rows, int h, int H0, void* stream) {
  SKG_REQUIRE(taps && ntaps > 0 && ntaps <= MAX_TAPS && bias0 && noise && Z && rows > 0 && h > 0);
  SKG_REQUIRE(H0 % 4 == 0 && skg_aligned(Z, 8) && samples > 0 && rows % samples == 0);
  SKG_REQUIRE(Wextra ? (ldw % 8 == 0 && skg_aligned(Wextra, 16)) : (h % GT == 0 && H0 % GC == 0));
  TapArgs a{};
  a.n = ntaps;
  for (int i = 0; i < ntaps; ++i) {
    SKG_REQUIRE(taps[i].P && taps[i].s > 0 && h % taps[i].s == 0 && skg_aligned(taps[i].P, 16));
    a.P[i] = taps[i].P;
    a.s[i] = taps[i].s;
  }
  if (h % GT == 0 && H0 % GC == 0) {
    hipLaunchKernelGGL(lgp_gather_tiled_kernel, dim3((h / GT) * (h / GT), rows, H0 / GC), dim3(256), 0,
                       (hipStream_t)stream, a, (const half_t*)Wextra, ldw, (const half_t*)bias0, noise, sigma, samples,
                       (half_t*)Z, rows, h, H0);
    SKG_CHECK_LAUNCH("skg_lgp_layer0_gather");
    return SKG_OK;
  }
  const size_t pixels = (size_t)rows * h * h;
  hipLaunchKernelGGL(lgp_gather_kernel, dim3((unsigned)((pixels + 1) / 2)), dim3(256), 0, (hipStream_t)stream, a,
                     (const half_t*)Wextra, ldw, (const half_t*)bias0, noise, sigma, samples, (half_t*)Z, rows, h, H0);
  SKG_CHECK_LAUNCH("skg_lgp_layer0_gather");
  return SKG_OK;
}

extern "C" int skg_lgp_layer0_scatter(const void* dZ, int lddz, void* dP, int rows, int h, int s, int H0,
                                      void* stream) {
  SKG_REQUIRE(dZ && dP && rows > 0 && h > 0 && s > 0 && h % s == 0 && H0 % 8 == 0 && lddz % 8 == 0 &&
              skg_aligned(dZ, 16) && skg_aligned(dP, 16));
  hipLaunchKernelGGL(lgp_scatter_kernel, dim3(rows * s * s), dim3(256), 0, (hipStream_t)stream, (const half_t*)dZ,
                     lddz, (half_t*)dP, rows, h, s, H0);
  SKG_CHECK_LAUNCH("skg_lgp_layer0_scatter");
  return SKG_OK;
}

extern "C" size_t skg_bn_scratch_floats(int samples, int C) {
  return (size_t)samples * BN_CHUNKS * C * 2 + (size_t)samples * C * 2;
}

extern "C" int skg_bn_stats(const void* X, int ldx, int samples, int segs, int seg_rows, int C, float eps,
                            float* stats, float* scratch, float* running_mean, float* running_var, void* stream) {
  SKG_REQUIRE(X && stats && scratch && samples > 0 && segs > 0 && seg_rows > 0 && C % 8 == 0 && ldx % 8 == 0 &&
              skg_aligned(X, 16));
  SKG_REQUIRE((running_mean == nullptr) == (running_var == nullptr));
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL((bn_partial_kernel<0>), dim3(BN_CHUNKS, samples), dim3(256), 0, st, (const half_t*)X, ldx,
                     (const half_t*)nullptr, 0, samples, segs, seg_rows, C, (const float*)nullptr, scratch);
  float* var = scratch + (size_t)samples * BN_CHUNKS * C * 2;      // [samples][C] (the "sums" area of the backward)
  hipLaunchKernelGGL((bn_finalize_kernel<0>), dim3(skg_cdiv(C, 128), samples), dim3(128), 0, st, scratch, samples, C,
                     segs * seg_rows, eps, stats, running_mean ? var : (float*)nullptr);
  if (running_mean)
    hipLaunchKernelGGL(bn_running_kernel, dim3(skg_cdiv(C, 128)), dim3(128), 0, st, stats, var, samples, C,
                       segs * seg_rows, running_mean, running_var);
  SKG_CHECK_LAUNCH("skg_bn_stats");
  return SKG_OK;
}

extern "C" int skg_bn_stats_from_running(const float* running_mean, const float* running_var, int samples, int C,
                                         float eps, float* stats, void* stream) {
  SKG_REQUIRE(running_mean && running_var && stats && samples > 0 && C > 0);
  hipLaunchKernelGGL(bn_from_running_kernel, dim3(skg_cdiv(samples * C, 256)), dim3(256), 0, (hipStream_t)stream,
                     running_mean, running_var, samples, C, eps, stats);
  SKG_CHECK_LAUNCH("skg_bn_stats_from_running");
  return SKG_OK;
}

extern "C" int skg_bn_apply(const void* X, int ldx, void* Y, int ldy, int samples, int segs, int seg_rows, int C,
                            const float* stats, const void* gamma, const void* beta, void* stream) {
  SKG_REQUIRE(X && Y && stats && gamma && beta && samples > 0 && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0);
  SKG_REQUIRE(skg_aligned(X, 16) && skg_aligned(Y, 16) && skg_aligned(gamma, 16) && skg_aligned(beta, 16));
  SKG_REQUIRE(segs > 0 && seg_rows > 0);
  hipLaunchKernelGGL((bn_apply_kernel<0>), dim3(bn_apply_chunks(segs * seg_rows, C), samples), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)X, ldx, (const half_t*)nullptr, 0, (half_t*)Y, ldy, samples, segs, seg_rows, C,
                     stats, (const float*)nullptr, (const half_t*)gamma, (const half_t*)beta, 1);
  SKG_CHECK_LAUNCH("skg_bn_apply");
  return SKG_OK;
}

extern "C" int skg_bn_relu_bwd(const void* X, int ldx, const void* dY, int lddy, void* dX, int lddx, int samples,
                               int segs, int seg_rows, int C, const float* stats, const void* gamma,
                               int train_mode, float* scratch, void* stream) {
  SKG_REQUIRE(X && dY && dX && stats && gamma && scratch && samples > 0 && C % 8 == 0);
  SKG_REQUIRE(ldx % 8 == 0 && lddy % 8 == 0 && lddx % 8 == 0 && skg_aligned(X, 16) && skg_aligned(dY, 16) &&
              skg_aligned(dX, 16) && skg_aligned(gamma, 16));
  hipStream_t st = (hipStream_t)stream;
  float* sums = scratch + (size_t)samples * BN_CHUNKS * C * 2;
  if (train_mode) {
    hipLaunchKernelGGL((bn_partial_kernel<1>), dim3(BN_CHUNKS, samples), dim3(256), 0, st, (const half_t*)X, ldx,
                       (const half_t*)dY, lddy, samples, segs, seg_rows, C, stats, scratch);
    hipLaunchKernelGGL((bn_finalize_kernel<1>), dim3(skg_cdiv(C, 128), samples), dim3(128), 0, st, scratch, samples, C,
                       segs * seg_rows, 0.f, sums, (float*)nullptr);
  }
  hipLaunchKernelGGL((bn_apply_kernel<1>), dim3(bn_apply_chunks(segs * seg_rows, C), samples), dim3(256), 0, st, (const half_t*)X, ldx,
                     (const half_t*)dY, lddy, (half_t*)dX, lddx, samples, segs, seg_rows, C, stats, sums,
                     (const half_t*)gamma, (const half_t*)nullptr, train_mode);
  SKG_CHECK_LAUNCH("skg_bn_relu_bwd");
  return SKG_OK;
}

extern "C" int skg_lgp_mse_seed(const void* out, int ldo, const float* target, void* dOut, int ldd, float* loss,
                                int samples, int h, float loss_scale, void* stream) {
  SKG_REQUIRE(out && target && dOut && samples > 0 && h > 0 && ldo >= 4 && ldd >= 4 && ldo % 4 == 0 && ldd % 4 == 0);
  SKG_REQUIRE(skg_aligned(out, 8) && skg_aligned(dOut, 8));
  if (hipMemsetAsync(dOut, 0, (size_t)2 * samples * h * h * ldd * sizeof(half_t), (hipStream_t)stream) != hipSuccess)
    return SKG_E_LAUNCH;
  hipLaunchKernelGGL(mse_seed_kernel, dim3(samples), dim3(256), 0, (hipStream_t)stream, (const half_t*)out, ldo,
                     target, (half_t*)dOut, ldd, loss, samples, h, loss_scale);
  SKG_CHECK_LAUNCH("skg_lgp_mse_seed");
  return SKG_OK;
}

// ---- training entry points ----------------------------------------------------------------------------------------
extern "C" size_t skg_colsum_scratch_floats(int C) { return (size_t)CS_CHUNKS * C; }

extern "C" int skg_colsum_f16(const void* X, int ldx, int M, int C, float scale, float* out, float* scratch,
                              void* stream) {
  SKG_REQUIRE(X && out && scratch && M > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldx >= C && skg_aligned(X, 16));
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL(colsum_partial_kernel, dim3(C / 8, CS_CHUNKS), dim3(256), 0, st, (const half_t*)X, ldx, M, C,
                     scratch);
  hipLaunchKernelGGL(colsum_final_kernel, dim3(skg_cdiv(C, 128)), dim3(128), 0, st, scratch, C, scale, out);
  SKG_CHECK_LAUNCH("skg_colsum_f16");
  return SKG_OK;
}

extern "C" int skg_bn_param_grads(const void* X, int ldx, const void* dY, int lddy, int rows, int C,
                                  const float* stats, float scale, float* dgamma, float* dbeta, float* scratch,
                                  void* stream) {
  SKG_REQUIRE(X && dY && stats && dgamma && dbeta && scratch && rows > 0 && C % 8 == 0 && ldx % 8 == 0 && lddy % 8 == 0 &&
              skg_aligned(X, 16) && skg_aligned(dY, 16));
  hipStream_t st = (hipStream_t)stream;
  hipLaunchKernelGGL((bn_partial_kernel<1>), dim3(BN_CHUNKS, 1), dim3(256), 0, st, (const half_t*)X, ldx,
                     (const half_t*)dY, lddy, 1, 1, rows, C, stats, scratch);
  hipLaunchKernelGGL(bn_param_grads_kernel, dim3(skg_cdiv(C, 128)), dim3(128), 0, st, scratch, C, scale, dgamma, dbeta);
  SKG_CHECK_LAUNCH("skg_bn_param_grads");
  return SKG_OK;
}

extern "C" int skg_lgp_extra_features(const float* noise, float sigma, int samples, int rows, int h, void* E, int ld,
                                      void* stream) {
  SKG_REQUIRE(noise && E && samples > 0 && rows > 0 && rows % samples == 0 && h > 0 && ld >= NEXTRA);
  hipLaunchKernelGGL(lgp_extra_kernel, dim3(ew_grid((size_t)rows * h * h * ld)), dim3(256), 0, (hipStream_t)stream,
                     noise, sigma, samples, rows, h, (half_t*)E, ld);
  SKG_CHECK_LAUNCH("skg_lgp_extra_features");
  return SKG_OK;
}

extern "C" int skg_lgp_mse_train(const void* out, int ldo, const float* target, void* dOut, int ldd,
                                 float* loss_parts, int samples, int h, float loss_scale, void* stream) {
  SKG_REQUIRE(out && target && dOut && loss_parts && samples > 0 && h > 0 && ldo >= 4 && ldd >= 4 && ldo % 4 == 0 &&
              ldd % 4 == 0 && skg_aligned(out, 8) && skg_aligned(dOut, 8));
  if (hipMemsetAsync(dOut, 0, (size_t)samples * h * h * ldd * sizeof(half_t), (hipStream_t)stream) != hipSuccess)
    return SKG_E_LAUNCH;
  hipLaunchKernelGGL(mse_train_kernel, dim3(samples), dim3(256), 0, (hipStream_t)stream, (const half_t*)out, ldo,
                     target, (half_t*)dOut, ldd, loss_parts, samples, h, loss_scale);
  SKG_CHECK_LAUNCH("skg_lgp_mse_train");
  return SKG_OK;
}

extern "C" int skg_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, void* param_f16,
                              size_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                              float inv_grad_scale, void* stream) {
  SKG_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step >= 1 && lr >= 0.f);
  const float bc1 = 1.f - powf(beta1, (float)step), bc2 = 1.f - powf(beta2, (float)step);
  hipLaunchKernelGGL(adamw_kernel, dim3(ew_grid(n)), dim3(256), 0, (hipStream_t)stream, param, grad, exp_avg,
                     exp_avg_sq, (half_t*)param_f16, n, lr, beta1, beta2, eps, weight_decay, bc1, bc2, inv_grad_scale);
  SKG_CHECK_LAUNCH("skg_adamw_step");
  return SKG_OK;
}

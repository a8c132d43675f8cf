// HBM-bound data-movement / pointwise kernels: GEGLU, transpose, axpby (concat / slice / grad
// accumulation), 2x2 sum-pool (adjoint of nearest upsample), NCHW<->NHWC casts, and the sampler's
// CFG + DDIM step and guidance update.  16-byte accesses, grid-stride loops.
#include "common.h"

namespace {

inline int ew_grid(size_t total_items) {
  size_t b = (total_items + 255) / 256;
  if (b > 4096) b = 4096;
  if (b < 1) b = 1;
  return (int)b;
}

// IL = 0: H = [a (F cols) | g (F cols)];  IL = 1: interleaved groups of four columns [a a g g] (the FF1 pack
// the fused GEMM epilogue uses), so both layouts produce the same Y.
template <int IL>
__global__ __launch_bounds__(256) void geglu_fwd_kernel(const half_t* __restrict__ H, int ldh,
                                                        half_t* __restrict__ Y, int ldy, int M, int F) {
  const int F8 = F >> 3;
  const size_t total = (size_t)M * F8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t m = i / F8;
    const int c = (int)(i - m * F8) * 8;
    half8_t a, g;
    if (IL) {
      const half8_t h0 = ld_half8(H + m * ldh + 2 * c), h1 = ld_half8(H + m * ldh + 2 * c + 8);
      a = half8_t{h0[0], h0[1], h0[4], h0[5], h1[0], h1[1], h1[4], h1[5]};
      g = half8_t{h0[2], h0[3], h0[6], h0[7], h1[2], h1[3], h1[6], h1[7]};
    } else {
      a = ld_half8(H + m * ldh + c);
      g = ld_half8(H + m * ldh + F + c);
    }
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)a[j] * gelu_f((float)g[j]));
    st_half8(Y + m * ldy + c, o);
  }
}

template <int IL>
__global__ __launch_bounds__(256) void geglu_bwd_kernel(const half_t* __restrict__ H, int ldh,
                                                        const half_t* __restrict__ dY, int lddy,
                                                        half_t* __restrict__ dH, int lddh, int M, int F) {
  const int F8 = F >> 3;
  const size_t total = (size_t)M * F8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t m = i / F8;
    const int c = (int)(i - m * F8) * 8;
    half8_t a, g;
    if (IL) {
      const half8_t h0 = ld_half8(H + m * ldh + 2 * c), h1 = ld_half8(H + m * ldh + 2 * c + 8);
      a = half8_t{h0[0], h0[1], h0[4], h0[5], h1[0], h1[1], h1[4], h1[5]};
      g = half8_t{h0[2], h0[3], h0[6], h0[7], h1[2], h1[3], h1[6], h1[7]};
    } else {
      a = ld_half8(H + m * ldh + c);
      g = ld_half8(H + m * ldh + F + c);
    }
    const half8_t d = ld_half8(dY + m * lddy + c);
    half8_t da, dg;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float gf = (float)g[j], df = (float)d[j];
      da[j] = (half_t)(df * gelu_f(gf));
      dg[j] = (half_t)(df * (float)a[j] * gelu_grad_f(gf));
    }
    if (IL) {
      st_half8(dH + m * lddh + 2 * c, half8_t{da[0], da[1], dg[0], dg[1], da[2], da[3], dg[2], dg[3]});
      st_half8(dH + m * lddh + 2 * c + 8, half8_t{da[4], da[5], dg[4], dg[5], da[6], da[7], dg[6], dg[7]});
    } else {
      st_half8(dH + m * lddh + c, da);
      st_half8(dH + m * lddh + F + c, dg);
    }
  }
}

// Out[c][m] = In[m][c]: 64x64 tile through LDS (pitch 66 halves -> conflict-light column reads).
__global__ __launch_bounds__(256) void transpose_kernel(const half_t* __restrict__ In, int ldi,
                                                        half_t* __restrict__ Out, int ldo, int M, int C) {
  __shared__ half_t tile[64][66];
  const int m0 = blockIdx.x * 64, c0 = blockIdx.y * 64;
  for (int pi = threadIdx.x; pi < 512; pi += 256) {
    const int r = pi >> 3, pc = (pi & 7) * 8;
    half8_t v = zero_half8();
    if (m0 + r < M && c0 + pc < C) v = ld_half8(In + (size_t)(m0 + r) * ldi + c0 + pc);
#pragma unroll
    for (int j = 0; j < 8; ++j) tile[r][pc + j] = v[j];
  }
  __syncthreads();
  for (int pi = threadIdx.x; pi < 512; pi += 256) {
    const int c = pi >> 3, pm = (pi & 7) * 8;
    if (c0 + c < C && m0 + pm < M) {
      half8_t v;
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = tile[pm + j][c];
      st_half8(Out + (size_t)(c0 + c) * ldo + m0 + pm, v);
    }
  }
}

__global__ __launch_bounds__(256) void axpby_kernel(const half_t* __restrict__ A, int lda,
                                                    const half_t* __restrict__ B, int ldb,
                                                    half_t* __restrict__ Y, int ldy, int M, int C, float alpha,
                                                    float beta) {
  const int C8 = C >> 3;
  const size_t total = (size_t)M * C8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t m = i / C8;
    const int c = (int)(i - m * C8) * 8;
    const half8_t a = ld_half8(A + m * lda + c);
    half8_t o;
    if (B) {
      const half8_t b = ld_half8(B + m * ldb + c);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (half_t)(alpha * (float)a[j] + beta * (float)b[j]);
    } else if (alpha == 1.f) {
      o = a;
    } else {
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = (half_t)(alpha * (float)a[j]);
    }
    st_half8(Y + m * ldy + c, o);
  }
}

// Out[b][r][:] = In[b][r][:] for r < rows_per_batch, with different batch strides on the two sides
__global__ __launch_bounds__(256) void batch_copy_kernel(const half_t* __restrict__ In, int ldi, long in_bs,
                                                         half_t* __restrict__ Out, int ldo, long out_bs,
                                                         int batches, int rows_per_batch, int C) {
  const int C8 = C >> 3;
  const size_t total = (size_t)batches * rows_per_batch * C8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % C8) * 8;
    const size_t t = i / C8;
    const int r = (int)(t % rows_per_batch);
    const size_t b = t / rows_per_batch;
    st_half8(Out + (b * out_bs + r) * ldo + c, ld_half8(In + (b * in_bs + r) * ldi + c));
  }
}

// quick_gelu(x) = x * sigmoid(1.702 x): the CLIP vision tower's MLP activation (transformers "quick_gelu")
__global__ __launch_bounds__(256) void quick_gelu_kernel(const half_t* __restrict__ X, int ldx, half_t* __restrict__ Y,
                                                         int ldy, int M, int C) {
  const int C8 = C >> 3;
  const size_t total = (size_t)M * C8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t m = i / C8;
    const int c = (int)(i - m * C8) * 8;
    const half8_t a = ld_half8(X + m * ldx + c);
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = (float)a[j];
      o[j] = (half_t)(x / (1.f + __expf(-1.702f * x)));
    }
    st_half8(Y + m * ldy + c, o);
  }
}

// gelu(x) = x/2 (1 + erf(x / sqrt 2)): the MLP activation of the OpenCLIP text encoder SD 2.x ships (hidden_act "gelu")
__global__ __launch_bounds__(256) void gelu_kernel(const half_t* __restrict__ X, int ldx, half_t* __restrict__ Y,
                                                   int ldy, int M, int C) {
  const int C8 = C >> 3;
  const size_t total = (size_t)M * C8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t m = i / C8;
    const int c = (int)(i - m * C8) * 8;
    const half8_t a = ld_half8(X + m * ldx + c);
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x = (float)a[j];
      o[j] = (half_t)(0.5f * x * (1.f + erff(x * 0.70710678118654752f)));
    }
    st_half8(Y + m * ldy + c, o);
  }
}

__global__ __launch_bounds__(256) void silu_kernel(const half_t* __restrict__ X, int ldx, half_t* __restrict__ Y,
                                                   int ldy, int M, int C) {
  const int C8 = C >> 3;
  const size_t total = (size_t)M * C8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t m = i / C8;
    const int c = (int)(i - m * C8) * 8;
    const half8_t a = ld_half8(X + m * ldx + c);
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)silu_f((float)a[j]);
    st_half8(Y + m * ldy + c, o);
  }
}

__global__ __launch_bounds__(256) void sumpool_kernel(const half_t* __restrict__ X, int ldx,
                                                      half_t* __restrict__ Y, int ldy, int rows, int H, int W,
                                                      int C) {
  const int C8 = C >> 3;
  const size_t total = (size_t)rows * H * W * C8;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t m = i / C8;
    const int c = (int)(i - m * C8) * 8;
    const int x = (int)(m % W);
    const size_t t = m / W;
    const int y = (int)(t % H);
    const size_t b = t / H;
    const size_t base = ((b * 2 * H + 2 * y) * 2 * W + 2 * x);
    const half8_t v0 = ld_half8(X + base * ldx + c);
    const half8_t v1 = ld_half8(X + (base + 1) * ldx + c);
    const half8_t v2 = ld_half8(X + (base + 2 * W) * ldx + c);
    const half8_t v3 = ld_half8(X + (base + 2 * W + 1) * ldx + c);
    half8_t o;
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)v0[j] + (float)v1[j] + (float)v2[j] + (float)v3[j]);
    st_half8(Y + m * ldy + c, o);
  }
}

__global__ __launch_bounds__(256) void nchw2nhwc_kernel(const float* __restrict__ X, half_t* __restrict__ Y,
                                                        int rows, int C, int HW, int Cpad) {
  const size_t total = (size_t)rows * HW * Cpad;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int c = (int)(i % Cpad);
    const size_t m = i / Cpad;
    const size_t b = m / HW;
    const int p = (int)(m - b * HW);
    Y[i] = c < C ? (half_t)X[(b * C + c) * HW + p] : (half_t)0.f;
  }
}

__global__ __launch_bounds__(256) void nhwc2nchw_kernel(const half_t* __restrict__ X, int ldx,
                                                        float* __restrict__ Y, int rows, int C, int HW) {
  const size_t total = (size_t)rows * C * HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int p = (int)(i % HW);
    const size_t t = i / HW;
    const int c = (int)(t % C);
    const size_t b = t / C;
    Y[i] = (float)X[(b * HW + p) * ldx + c];
  }
}

// CFG combine + DDIM (eta = 0) on float NCHW latents, 4 channels.
// lo_off != 0 (accuracy mode): eps is a PAIR, the lo part lo_off columns to the right of the hi part in the same rows
// vpred: the model output is v (prediction_type "v_prediction", SD2.1-768): eps = c0 v + c1 x, x0 = c0 x - c1 v
__global__ __launch_bounds__(256) void cfg_ddim_kernel(const half_t* __restrict__ eu, const half_t* __restrict__ ec,
                                                       int ld, int lo_off, const float* __restrict__ x,
                                                       float* __restrict__ xp, float* __restrict__ eps_out,
                                                       int samples, int HW, float g, float c0, float c1, float c2,
                                                       float c3, int vpred) {
  const size_t total = (size_t)samples * 4 * HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int p = (int)(i % HW);
    const size_t t = i / HW;
    const int c = (int)(t & 3);
    const size_t s = t >> 2;
    const size_t off = (s * HW + p) * ld + c;
    float u = (float)eu[off], v = (float)ec[off];
    if (lo_off) { u += (float)eu[off + lo_off]; v += (float)ec[off + lo_off]; }
    float e = u + g * (v - u);
    float x0;
    if (vpred) {
      const float xv = x[i], vv = e;
      e = c0 * vv + c1 * xv;
      x0 = c0 * xv - c1 * vv;
    } else {
      x0 = (x[i] - c1 * e) / c0;
    }
    xp[i] = c2 * x0 + c3 * e;
    if (eps_out) eps_out[i] = e;
  }
}

// Row softmax of fp16 scores (one workgroup per row, fp32 arithmetic, fp16 probabilities): the single-head
// attention of the VAE decoder (N = HW = 4096 keys, head width 512) runs as GEMM -> softmax -> GEMM.
__global__ __launch_bounds__(256) void softmax_rows_kernel(const half_t* __restrict__ x, int ldx,
                                                           half_t* __restrict__ y, int ldy, int N) {
  __shared__ float red[8];
  const size_t row = blockIdx.x;
  const half_t* xr = x + row * ldx;
  half_t* yr = y + row * ldy;
  float m = -3.0e38f;
  for (int i = threadIdx.x * 8; i < N; i += 256 * 8) {
    const half8_t v = ld_half8(xr + i);
#pragma unroll
    for (int e = 0; e < 8; ++e) m = fmaxf(m, (float)v[e]);
  }
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  float sum = 0.f;
  for (int i = threadIdx.x * 8; i < N; i += 256 * 8) {
    const half8_t v = ld_half8(xr + i);
#pragma unroll
    for (int e = 0; e < 8; ++e) sum += __expf((float)v[e] - m);
  }
  sum = block_sum<256>(sum, red + 4);
  const float inv = 1.f / sum;
  for (int i = threadIdx.x * 8; i < N; i += 256 * 8) {
    const half8_t v = ld_half8(xr + i);
    half8_t o;
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = (half_t)(__expf((float)v[e] - m) * inv);
    st_half8(yr + i, o);
  }
}

// decode_latents tail: fp16 NHWC [pixels][ld] (first C channels) -> float NHWC [pixels][C] = clamp(x*scale + shift, 0, 1)
__global__ __launch_bounds__(256) void image_post_kernel(const half_t* __restrict__ x, int ld, float* __restrict__ out,
                                                         size_t pixels, int C, float scale, float shift) {
  const size_t total = pixels * C;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const size_t px = i / C;
    const int c = (int)(i - px * C);
    out[i] = fminf(fmaxf((float)x[px * ld + c] * scale + shift, 0.f), 1.f);
  }
}

// decode_latents tail + numpy_to_pil's quantisation (modules/pipeline.py:118,125): one thread per pixel,
// u8[px][c] = rint(clamp(x*scale + shift, 0, 1) * 255)  (numpy's round = half to even = v_rndne)
__global__ __launch_bounds__(256) void image_u8_kernel(const half_t* __restrict__ x, int ld, unsigned char* __restrict__ out,
                                                       size_t pixels, int C, float scale, float shift) {
  for (size_t px = (size_t)blockIdx.x * 256 + threadIdx.x; px < pixels; px += (size_t)gridDim.x * 256) {
    const half_t* src = x + px * ld;
    unsigned char* dst = out + px * C;
    for (int c = 0; c < C; ++c)
      dst[c] = (unsigned char)rintf(fminf(fmaxf((float)src[c] * scale + shift, 0.f), 1.f) * 255.f);
  }
}

// DiagonalGaussianDistribution.sample() * scale from the encoder's moments: fp16 NHWC [px][ld] = (mean[0..L), logvar[L..2L))
// -> float NCHW [S][L][HW]; logvar clamped to [-30, 20]; noise = caller-drawn N(0,1), NCHW (NULL: the mode = mean)
__global__ __launch_bounds__(256) void gaussian_sample_kernel(const half_t* __restrict__ m, int ld,
                                                              const float* __restrict__ noise, float* __restrict__ out,
                                                              int S, int L, int HW, float scale) {
  const size_t total = (size_t)S * L * HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int p = (int)(i % HW);
    const size_t t = i / HW;
    const int c = (int)(t % L);
    const size_t s = t / L;
    const half_t* px = m + (s * HW + p) * ld;
    float v = (float)px[c];
    if (noise) {
      const float lv = fminf(fmaxf((float)px[L + c], -30.f), 20.f);
      v += __expf(0.5f * lv) * noise[i];
    }
    out[i] = v * scale;
  }
}

// CFG combine + one DPM-Solver++ (2M) update: x0 = (x - sigma_s*eps)/alpha_s; x_prev = a*x + b*x0 + c*x0_before.
// x0_io holds the previous step's x0 on entry (ignored when c == 0) and this step's x0 on exit.
// vpred: the model output is v: x0 = alpha_s x - sigma_s v (eps_out: alpha_s v + sigma_s x)
__global__ __launch_bounds__(256) void cfg_dpm_kernel(const half_t* __restrict__ eu, const half_t* __restrict__ ec,
                                                      int ld, int lo_off, const float* __restrict__ x, float* __restrict__ x0_io,
                                                      float* __restrict__ xp, float* __restrict__ eps_out,
                                                      int samples, int HW, float g, float alpha_s, float sigma_s,
                                                      float a, float b, float c, int vpred) {
  const size_t total = (size_t)samples * 4 * HW;
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
    const int p = (int)(i % HW);
    const size_t t = i / HW;
    const int ch = (int)(t & 3);
    const size_t s = t >> 2;
    const size_t off = (s * HW + p) * ld + ch;
    float u = (float)eu[off], v = (float)ec[off];
    if (lo_off) { u += (float)eu[off + lo_off]; v += (float)ec[off + lo_off]; }
    float e = u + g * (v - u);
    const float xv = x[i];
    float x0;
    if (vpred) {
      x0 = alpha_s * xv - sigma_s * e;
      e = alpha_s * e + sigma_s * xv;
    } else {
      x0 = (xv - sigma_s * e) / alpha_s;
    }
    float r = a * xv + b * x0;
    if (c != 0.f) r += c * x0_io[i];
    x0_io[i] = x0;
    xp[i] = r;
    if (eps_out) eps_out[i] = e;
  }
}

// one block per sample: alpha = sqrt(2)*||x_in - x_prev|| / ||g|| * beta; x_prev += alpha*g, g = -grad
__global__ __launch_bounds__(256) void guidance_update_kernel(const half_t* __restrict__ grad, int ld,
                                                              const float* __restrict__ x_in,
                                                              float* __restrict__ x_prev, float* __restrict__ aux,
                                                              int HW, float beta) {
  __shared__ float red[8];
  const int s = blockIdx.x;
  const int n = 4 * HW;
  const float* xi = x_in + (size_t)s * n;
  float* xp = x_prev + (size_t)s * n;
  const half_t* gr = grad + (size_t)s * HW * ld;
  float sd = 0.f, sg = 0.f;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int c = i / HW, p = i - c * HW;
    const float d = xi[i] - xp[i];
    const float gv = (float)gr[(size_t)p * ld + c];
    sd += d * d;
    sg += gv * gv;
  }
  sd = block_sum<256>(sd, red);
  sg = block_sum<256>(sg, red);
  const float num = sqrtf(2.f * sd), den = sqrtf(sg);
  const float alpha = num / den * beta;
  for (int i = threadIdx.x; i < n; i += 256) {
    const int c = i / HW, p = i - c * HW;
    xp[i] -= alpha * (float)gr[(size_t)p * ld + c];
  }
  if (threadIdx.x == 0 && aux) {
    aux[s * 4 + 0] = alpha; aux[s * 4 + 1] = den; aux[s * 4 + 2] = num; aux[s * 4 + 3] = 0.f;
  }
}

}  // namespace

extern "C" int skg_geglu_fwd(const void* H, int ldh, void* Y, int ldy, int M, int F, int interleaved,
                             void* stream) {
  SKG_REQUIRE(H && Y && M > 0 && F > 0 && F % 8 == 0 && ldh % 8 == 0 && ldy % 8 == 0 && ldh >= 2 * F);
  SKG_REQUIRE(skg_aligned(H, 16) && skg_aligned(Y, 16));
  if (interleaved)
    hipLaunchKernelGGL((geglu_fwd_kernel<1>), dim3(ew_grid((size_t)M * F / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)H, ldh, (half_t*)Y, ldy, M, F);
  else
    hipLaunchKernelGGL((geglu_fwd_kernel<0>), dim3(ew_grid((size_t)M * F / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)H, ldh, (half_t*)Y, ldy, M, F);
  SKG_CHECK_LAUNCH("skg_geglu_fwd");
  return SKG_OK;
}

extern "C" int skg_geglu_bwd(const void* H, int ldh, const void* dY, int lddy, void* dH, int lddh, int M,
                             int F, int interleaved, void* stream) {
  SKG_REQUIRE(H && dY && dH && M > 0 && F % 8 == 0 && ldh % 8 == 0 && lddy % 8 == 0 && lddh % 8 == 0);
  SKG_REQUIRE(skg_aligned(H, 16) && skg_aligned(dY, 16) && skg_aligned(dH, 16));
  if (interleaved)
    hipLaunchKernelGGL((geglu_bwd_kernel<1>), dim3(ew_grid((size_t)M * F / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)H, ldh, (const half_t*)dY, lddy, (half_t*)dH, lddh, M, F);
  else
    hipLaunchKernelGGL((geglu_bwd_kernel<0>), dim3(ew_grid((size_t)M * F / 8)), dim3(256), 0, (hipStream_t)stream,
                       (const half_t*)H, ldh, (const half_t*)dY, lddy, (half_t*)dH, lddh, M, F);
  SKG_CHECK_LAUNCH("skg_geglu_bwd");
  return SKG_OK;
}

extern "C" int skg_transpose_f16(const void* In, int ldi, void* Out, int ldo, int M, int C, void* stream) {
  SKG_REQUIRE(In && Out && M > 0 && C > 0 && M % 8 == 0 && C % 8 == 0 && ldi % 8 == 0 && ldo % 8 == 0);
  SKG_REQUIRE(skg_aligned(In, 16) && skg_aligned(Out, 16) && ldi >= C && ldo >= M);
  hipLaunchKernelGGL(transpose_kernel, dim3(skg_cdiv(M, 64), skg_cdiv(C, 64)), dim3(256), 0,
                     (hipStream_t)stream, (const half_t*)In, ldi, (half_t*)Out, ldo, M, C);
  SKG_CHECK_LAUNCH("skg_transpose_f16");
  return SKG_OK;
}

extern "C" int skg_axpby_f16(const void* A, int lda, const void* B, int ldb, void* Y, int ldy, int M, int C,
                             float alpha, float beta, void* stream) {
  SKG_REQUIRE(A && Y && M > 0 && C > 0 && C % 8 == 0 && lda % 8 == 0 && ldy % 8 == 0 && (!B || ldb % 8 == 0));
  SKG_REQUIRE(skg_aligned(A, 16) && skg_aligned(Y, 16) && (!B || skg_aligned(B, 16)));
  hipLaunchKernelGGL(axpby_kernel, dim3(ew_grid((size_t)M * C / 8)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)A, lda, (const half_t*)B, ldb, (half_t*)Y, ldy, M, C, alpha, beta);
  SKG_CHECK_LAUNCH("skg_axpby_f16");
  return SKG_OK;
}

extern "C" int skg_batch_copy_f16(const void* In, int ldi, int in_batch_rows, void* Out, int ldo,
                                  int out_batch_rows, int batches, int rows_per_batch, int C, void* stream) {
  SKG_REQUIRE(In && Out && batches > 0 && rows_per_batch > 0 && C > 0 && C % 8 == 0 && ldi % 8 == 0 && ldo % 8 == 0);
  SKG_REQUIRE(in_batch_rows >= rows_per_batch && out_batch_rows >= rows_per_batch);
  SKG_REQUIRE(skg_aligned(In, 16) && skg_aligned(Out, 16));
  hipLaunchKernelGGL(batch_copy_kernel, dim3(ew_grid((size_t)batches * rows_per_batch * C / 8)), dim3(256), 0,
                     (hipStream_t)stream, (const half_t*)In, ldi, (long)in_batch_rows, (half_t*)Out, ldo,
                     (long)out_batch_rows, batches, rows_per_batch, C);
  SKG_CHECK_LAUNCH("skg_batch_copy_f16");
  return SKG_OK;
}

extern "C" int skg_silu_f16(const void* X, int ldx, void* Y, int ldy, int M, int C, void* stream) {
  SKG_REQUIRE(X && Y && M > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0);
  SKG_REQUIRE(skg_aligned(X, 16) && skg_aligned(Y, 16));
  hipLaunchKernelGGL(silu_kernel, dim3(ew_grid((size_t)M * C / 8)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)X, ldx, (half_t*)Y, ldy, M, C);
  SKG_CHECK_LAUNCH("skg_silu_f16");
  return SKG_OK;
}

extern "C" int skg_quick_gelu_f16(const void* X, int ldx, void* Y, int ldy, int M, int C, void* stream) {
  SKG_REQUIRE(X && Y && M > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C);
  SKG_REQUIRE(skg_aligned(X, 16) && skg_aligned(Y, 16));
  hipLaunchKernelGGL(quick_gelu_kernel, dim3(ew_grid((size_t)M * C / 8)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)X, ldx, (half_t*)Y, ldy, M, C);
  SKG_CHECK_LAUNCH("skg_quick_gelu_f16");
  return SKG_OK;
}

extern "C" int skg_gelu_f16(const void* X, int ldx, void* Y, int ldy, int M, int C, void* stream) {
  SKG_REQUIRE(X && Y && M > 0 && C > 0 && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0 && ldx >= C && ldy >= C);
  SKG_REQUIRE(skg_aligned(X, 16) && skg_aligned(Y, 16));
  hipLaunchKernelGGL(gelu_kernel, dim3(ew_grid((size_t)M * C / 8)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)X, ldx, (half_t*)Y, ldy, M, C);
  SKG_CHECK_LAUNCH("skg_gelu_f16");
  return SKG_OK;
}

extern "C" int skg_sumpool2x2_f16(const void* X, int ldx, void* Y, int ldy, int rows, int H, int W, int C,
                                  void* stream) {
  SKG_REQUIRE(X && Y && rows > 0 && H > 0 && W > 0 && C % 8 == 0 && ldx % 8 == 0 && ldy % 8 == 0);
  SKG_REQUIRE(skg_aligned(X, 16) && skg_aligned(Y, 16));
  hipLaunchKernelGGL(sumpool_kernel, dim3(ew_grid((size_t)rows * H * W * C / 8)), dim3(256), 0,
                     (hipStream_t)stream, (const half_t*)X, ldx, (half_t*)Y, ldy, rows, H, W, C);
  SKG_CHECK_LAUNCH("skg_sumpool2x2_f16");
  return SKG_OK;
}

extern "C" int skg_nchw_f32_to_nhwc_f16(const float* X, void* Y, int rows, int C, int HW, int Cpad,
                                        void* stream) {
  SKG_REQUIRE(X && Y && rows > 0 && C > 0 && HW > 0 && Cpad >= C);
  hipLaunchKernelGGL(nchw2nhwc_kernel, dim3(ew_grid((size_t)rows * HW * Cpad)), dim3(256), 0,
                     (hipStream_t)stream, X, (half_t*)Y, rows, C, HW, Cpad);
  SKG_CHECK_LAUNCH("skg_nchw_f32_to_nhwc_f16");
  return SKG_OK;
}

extern "C" int skg_nhwc_f16_to_nchw_f32(const void* X, int ldx, float* Y, int rows, int C, int HW,
                                        void* stream) {
  SKG_REQUIRE(X && Y && rows > 0 && C > 0 && HW > 0 && ldx >= C);
  hipLaunchKernelGGL(nhwc2nchw_kernel, dim3(ew_grid((size_t)rows * HW * C)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)X, ldx, Y, rows, C, HW);
  SKG_CHECK_LAUNCH("skg_nhwc_f16_to_nchw_f32");
  return SKG_OK;
}

extern "C" int skg_cfg_ddim_step(const void* eps_u, const void* eps_c, int ld, int lo_off, const float* x, float* x_prev,
                                 float* eps_out, int samples, int HW, float g, float c0, float c1, float c2,
                                 float c3, int vpred, void* stream) {
  SKG_REQUIRE(eps_u && eps_c && x && x_prev && samples > 0 && HW > 0 && ld >= 4 && lo_off >= 0 && (lo_off == 0 || ld >= lo_off + 4));
  SKG_REQUIRE(vpred == 0 || vpred == 1);
  hipLaunchKernelGGL(cfg_ddim_kernel, dim3(ew_grid((size_t)samples * 4 * HW)), dim3(256), 0,
                     (hipStream_t)stream, (const half_t*)eps_u, (const half_t*)eps_c, ld, lo_off, x, x_prev, eps_out,
                     samples, HW, g, c0, c1, c2, c3, vpred);
  SKG_CHECK_LAUNCH("skg_cfg_ddim_step");
  return SKG_OK;
}

extern "C" int skg_softmax_rows_f16(const void* x, int ldx, void* y, int ldy, int M, int N, void* stream) {
  SKG_REQUIRE(x && y && M > 0 && N > 0 && N % 8 == 0 && ldx >= N && ldy >= N && ldx % 8 == 0 && ldy % 8 == 0);
  SKG_REQUIRE(skg_aligned(x, 16) && skg_aligned(y, 16));
  hipLaunchKernelGGL(softmax_rows_kernel, dim3(M), dim3(256), 0, (hipStream_t)stream, (const half_t*)x, ldx,
                     (half_t*)y, ldy, N);
  SKG_CHECK_LAUNCH("skg_softmax_rows_f16");
  return SKG_OK;
}

extern "C" int skg_image_postprocess(const void* x, int ld, float* out, size_t pixels, int C, float scale,
                                     float shift, void* stream) {
  SKG_REQUIRE(x && out && pixels > 0 && C > 0 && ld >= C);
  hipLaunchKernelGGL(image_post_kernel, dim3(ew_grid((size_t)pixels * C)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)x, ld, out, (size_t)pixels, C, scale, shift);
  SKG_CHECK_LAUNCH("skg_image_postprocess");
  return SKG_OK;
}

extern "C" int skg_image_to_u8(const void* x, int ld, void* out, size_t pixels, int C, float scale, float shift,
                               void* stream) {
  SKG_REQUIRE(x && out && pixels > 0 && C > 0 && ld >= C);
  hipLaunchKernelGGL(image_u8_kernel, dim3(ew_grid((size_t)pixels)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)x, ld, (unsigned char*)out, (size_t)pixels, C, scale, shift);
  SKG_CHECK_LAUNCH("skg_image_to_u8");
  return SKG_OK;
}

extern "C" int skg_gaussian_sample(const void* moments, int ld, const float* noise, float* out, int samples, int L,
                                   int HW, float scale, void* stream) {
  SKG_REQUIRE(moments && out && samples > 0 && L > 0 && HW > 0 && ld >= 2 * L);
  hipLaunchKernelGGL(gaussian_sample_kernel, dim3(ew_grid((size_t)samples * L * HW)), dim3(256), 0,
                     (hipStream_t)stream, (const half_t*)moments, ld, noise, out, samples, L, HW, scale);
  SKG_CHECK_LAUNCH("skg_gaussian_sample");
  return SKG_OK;
}

extern "C" int skg_cfg_dpmpp2m_step(const void* eps_u, const void* eps_c, int ld, int lo_off, const float* x, float* x0_io,
                                    float* x_prev, float* eps_out, int samples, int HW, float g, float alpha_s,
                                    float sigma_s, float a, float b, float c, int vpred, void* stream) {
  SKG_REQUIRE(eps_u && eps_c && x && x0_io && x_prev && samples > 0 && HW > 0 && ld >= 4 && alpha_s > 0.f);
  SKG_REQUIRE(vpred == 0 || vpred == 1);
  SKG_REQUIRE(lo_off >= 0 && (lo_off == 0 || ld >= lo_off + 4));
  hipLaunchKernelGGL(cfg_dpm_kernel, dim3(ew_grid((size_t)samples * 4 * HW)), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)eps_u, (const half_t*)eps_c, ld, lo_off, x, x0_io, x_prev, eps_out, samples, HW, g,
                     alpha_s, sigma_s, a, b, c, vpred);
  SKG_CHECK_LAUNCH("skg_cfg_dpmpp2m_step");
  return SKG_OK;
}

extern "C" int skg_guidance_update(const void* grad, int ld, const float* x_in, float* x_prev, float* aux,
                                   int samples, int HW, float beta, void* stream) {
  SKG_REQUIRE(grad && x_in && x_prev && samples > 0 && HW > 0 && ld >= 4);
  hipLaunchKernelGGL(guidance_update_kernel, dim3(samples), dim3(256), 0, (hipStream_t)stream,
                     (const half_t*)grad, ld, x_in, x_prev, aux, HW, beta);
  SKG_CHECK_LAUNCH("skg_guidance_update");
  return SKG_OK;
}

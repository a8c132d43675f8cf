// Shared device helpers for the skg kernels (gfx950 / CDNA4 only: 64-lane wavefronts).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/skg.h"

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));

#define SKG_WAVE 64

// host side ------------------------------------------------------------------------------------
void skg_set_error(const char* what, hipError_t e);
#define SKG_CHECK_LAUNCH(name)                                  \
  do {                                                          \
    hipError_t e__ = hipGetLastError();                         \
    if (e__ != hipSuccess) {                                    \
      skg_set_error(name, e__);                                 \
      return SKG_E_LAUNCH;                                      \
    }                                                           \
  } while (0)
#define SKG_REQUIRE(cond)               \
  do {                                  \
    if (!(cond)) return SKG_E_BADARG;   \
  } while (0)

static inline bool skg_aligned(const void* p, size_t a) { return (reinterpret_cast<uintptr_t>(p) % a) == 0; }
static inline int skg_cdiv(int a, int b) { return (a + b - 1) / b; }

// device side ----------------------------------------------------------------------------------
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// block-wide sum of `v` for blocks of NT threads (NT multiple of 64); result valid on all threads.
template <int NT>
__device__ __forceinline__ float block_sum(float v, float* smem /* >= NT/64 floats */) {
  v = wave_sum(v);
  const int w = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) smem[w] = v;
  __syncthreads();
  float r = 0.f;
#pragma unroll
  for (int i = 0; i < NT / 64; ++i) r += smem[i];
  return r;
}

__device__ __forceinline__ half8_t ld_half8(const half_t* p) { return *reinterpret_cast<const half8_t*>(p); }
__device__ __forceinline__ void st_half8(half_t* p, half8_t v) { *reinterpret_cast<half8_t*>(p) = v; }
__device__ __forceinline__ half4_t ld_half4(const half_t* p) { return *reinterpret_cast<const half4_t*>(p); }
__device__ __forceinline__ void st_half4(half_t* p, half4_t v) { *reinterpret_cast<half4_t*>(p) = v; }

__device__ __forceinline__ half8_t zero_half8() {
  half8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
  return z;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.f + __expf(-x)); }
__device__ __forceinline__ float silu_grad_f(float x) {
  const float s = 1.f / (1.f + __expf(-x));
  return s * (1.f + x * (1.f - s));
}
__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
// exact-erf GELU with erf from Abramowitz & Stegun 7.1.26 (|abs error| < 1.5e-7, far below the fp16 output
// rounding): one v_rcp + one v_exp + a 5-term Horner chain, ~half the instructions of erff().  Used where the
// GELU sits on a compute-bound path (fused GEMM epilogue).
__device__ __forceinline__ float gelu_fast_f(float x) {
  const float z = fabsf(x) * 0.70710678118654752f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.f));
  float p = fmaf(1.061405429f, t, -1.453152027f);
  p = fmaf(p, t, 1.421413741f);
  p = fmaf(p, t, -0.284496736f);
  p = fmaf(p, t, 0.254829592f);
  const float e = 1.f - p * t * __builtin_amdgcn_exp2f(-z * z * 1.4426950408889634f);   // erf(|x|/sqrt2)
  return 0.5f * x + 0.5f * fabsf(x) * e;            // 0.5*x*(1 + sign(x)*erf)
}
__device__ __forceinline__ float gelu_grad_f(float x) {
  const float cdf = 0.5f * (1.f + erff(x * 0.70710678118654752f));
  const float pdf = 0.39894228040143268f * __expf(-0.5f * x * x);
  return cdf + x * pdf;
}

// x + (x of the lane DPP control CTRL points at), inside a 16-lane row: one VALU instruction, no LDS (quad_perm
// 0xB1 = lane ^ 1, 0x4E = lane ^ 2; row_ror:4 = 0x124, row_ror:8 = 0x128 - rotations: fine for an all-reduce)
template <int CTRL>
__device__ __forceinline__ float dpp_add(float x) {
  const int y = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xf, 0xf, false);
  return x + __builtin_bit_cast(float, y);
}

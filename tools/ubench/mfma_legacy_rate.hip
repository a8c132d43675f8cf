// Issue rate of the CDNA3-era f16 MFMA shapes on gfx950 beside the doubled-K ones: does v_mfma_f32_16x16x16_f16 (half the flops
// of 16x16x32) take half the cycles?  If it did, a d = 40 head could run QK^T as one K = 32 step + one K = 16 step instead of
// two K = 32 steps (attention.hip pads 40 to 64).  One wave per SIMD, 4 independent accumulators, s_memtime around 4096 MFMAs.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_legacy_rate.hip -o tools/ubench/mfma_legacy_rate && tools/ubench/mfma_legacy_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));

template <int WHICH>
__global__ __launch_bounds__(256) void rate(unsigned long long* out, float* sink) {
  half8_t a8, b8;
  half4_t a4, b4;
  for (int j = 0; j < 8; ++j) { a8[j] = (_Float16)(threadIdx.x * 0.001f + j); b8[j] = (_Float16)(0.5f - j); }
  for (int j = 0; j < 4; ++j) { a4[j] = a8[j]; b4[j] = b8[j]; }
  float4_t c[4] = {};
  float16_t d[2] = {};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < 1024; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (WHICH == 0) c[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[u], 0, 0, 0);
      if (WHICH == 1) c[u] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c[u], 0, 0, 0);
      if (WHICH == 2) d[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a8, b8, d[u & 1], 0, 0, 0);
      if (WHICH == 3) d[u & 1] = __builtin_amdgcn_mfma_f32_32x32x8f16(a4, b4, d[u & 1], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0.f;
  for (int u = 0; u < 4; ++u) s += c[u][0] + c[u][3];
  s += d[0][0] + d[1][5];
  if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
  if (s == 12345.f) sink[0] = s;
}

int main() {
  unsigned long long* d; float* s;
  hipMalloc(&d, 256 * 8); hipMalloc(&s, 4);
  const char* names[4] = {"v_mfma_f32_16x16x32_f16", "v_mfma_f32_16x16x16_f16", "v_mfma_f32_32x32x16_f16", "v_mfma_f32_32x32x8_f16"};
  for (int w = 0; w < 4; ++w) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (w == 0) hipLaunchKernelGGL(rate<0>, dim3(256), dim3(256), 0, 0, d, s);
      if (w == 1) hipLaunchKernelGGL(rate<1>, dim3(256), dim3(256), 0, 0, d, s);
      if (w == 2) hipLaunchKernelGGL(rate<2>, dim3(256), dim3(256), 0, 0, d, s);
      if (w == 3) hipLaunchKernelGGL(rate<3>, dim3(256), dim3(256), 0, 0, d, s);
      hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[256]; hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    const double flops = (w == 1 || w == 3) ? 8192.0 * (w == 3 ? 2 : 1) : 16384.0 * (w == 2 ? 2 : 1);
    printf("%-26s %7.2f s_memtime ticks per MFMA (100 MHz counter), kernel %.3f ms -> %.1f ns per MFMA per SIMD, %.0f TFLOP/s chip\n", names[w],
           (double)h[0] / 4096.0, ms, ms * 1e6 / 4096.0, 256.0 * 4 * 4096 * flops / (ms * 1e-3) / 1e12);
  }
  return 0;
}

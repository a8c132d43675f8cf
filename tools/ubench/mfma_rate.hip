// Microbenchmark: issue rate of v_mfma_f32_16x16x32_f16 (gfx950) vs the legacy v_mfma_f32_16x16x16_f16 on one wave per
// SIMD and four independent accumulators.  hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_rate.hip -o tools/ubench/mfma_rate
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float float4_t __attribute__((ext_vector_type(4)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  half8_t a8, b8;
  half4_t a4, b4;
  for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(threadIdx.x * 0.001f + i); b8[i] = (_Float16)(0.5f + i); }
  for (int i = 0; i < 4; ++i) { a4[i] = a8[i]; b4[i] = b8[i]; }
  float4_t c[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (KIND == 0) c[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[u], 0, 0, 0);
      else c[u] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4, b4, c[u], 0, 0, 0);
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 256 + threadIdx.x] = c[0][0] + c[1][1] + c[2][2] + c[3][3];
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 256 * 256 * 4); hipMalloc(&cyc, 256 * 8);
  const int iters = 20000;
  for (int kind = 0; kind < 2; ++kind) {
    if (kind == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    else hipLaunchKernelGGL(k<1>, dim3(256), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[256];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256; ++i) avg += (double)h[i];
    avg /= 256;
    printf("%s: %.2f cycles per MFMA per wave (1 wave / SIMD, 4 accumulators)\n", kind == 0 ? "16x16x32_f16" : "16x16x16_f16", avg / (iters * 4.0));
  }
  return 0;
}

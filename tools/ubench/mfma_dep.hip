// Microbenchmark: v_mfma_f32_16x16x32_f16 cost as a function of the number of independent accumulators a wave cycles
// through (dependency distance) and of the waves per SIMD; ticks of s_memtime calibrated against wall clock.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_dep.hip -o tools/ubench/mfma_dep
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  half8_t a8, b8;
  for (int i = 0; i < 8; ++i) { a8[i] = (_Float16)(threadIdx.x * 0.001f + i); b8[i] = (_Float16)(0.5f + i); }
  float4_t c[NACC];
  for (int u = 0; u < NACC; ++u) c[u] = float4_t{0, 0, 0, 0};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < NACC; ++u) c[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a8, b8, c[u], 0, 0, 0);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int u = 0; u < NACC; ++u) s += c[u][0];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NACC> void run(float* out, unsigned long long* cyc) {
  const int iters = 16384 / NACC;
  for (int wps = 1; wps <= 4; wps *= 2) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<NACC>, dim3(256 * wps), dim3(256), 0, 0, out, cyc, iters);   // warm
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<NACC>, dim3(256 * wps), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024];
    hipMemcpy(h, cyc, sizeof(unsigned long long) * 256 * wps, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256 * wps; ++i) avg += (double)h[i];
    avg /= 256 * wps;
    const double n = (double)iters * NACC;          // MFMAs per wave
    printf("acc %2d  waves/SIMD %d: %6.1f ticks/MFMA/wave  %6.1f ticks/MFMA/SIMD   wall %.1f us -> %.2f ticks/ns, %6.1f TFLOP/s\n", NACC, wps,
           avg / n, avg / n / wps, ms * 1e3, avg / (ms * 1e6), n * 16384.0 * 1024 * wps / (ms * 1e-3) / 1e12);
  }
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
  run<1>(out, cyc); run<2>(out, cyc); run<4>(out, cyc); run<8>(out, cyc); run<16>(out, cyc);
  return 0;
}

// Microbenchmark: issue cost (cycles per wave-instruction, one wave per SIMD, 8 independent chains) of the VALU
// instructions the attention softmax is made of on gfx950.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/valu_rate.hip -o tools/ubench/valu_rate
#include <hip/hip_runtime.h>
#include <stdio.h>

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, unsigned long long* cyc, int iters) {
  float v[8];
  for (int i = 0; i < 8; ++i) v[i] = -0.001f * (threadIdx.x + 1) - i;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      if (KIND == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[u]));
      else if (KIND == 1) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[u]));
      else if (KIND == 2) asm volatile("v_exp_f16 %0, %0" : "+v"(v[u]));
      else if (KIND == 3) asm volatile("v_max3_f32 %0, %0, %0, %0" : "+v"(v[u]));
      else if (KIND == 4) asm volatile("v_cvt_pk_f16_f32 %0, %0, %0" : "+v"(v[u]));
      else if (KIND == 5) asm volatile("v_pk_mul_f32 %0, %0, %0" : "+v"(*reinterpret_cast<double*>(&v[u & ~1])));
      else if (KIND == 6) asm volatile("v_rcp_f32 %0, %0" : "+v"(v[u]));
    }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0; for (int i = 0; i < 8; ++i) s += v[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
  if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int KIND> void run(const char* name, float* out, unsigned long long* cyc) {
  const int iters = 20000;
  printf("%-18s", name);
  for (int wps = 1; wps <= 4; wps *= 2) {          // waves per SIMD (workgroups of 4 waves per CU)
    hipLaunchKernelGGL(k<KIND>, dim3(256 * wps), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    unsigned long long h[1024];
    hipMemcpy(h, cyc, sizeof(unsigned long long) * 256 * wps, hipMemcpyDeviceToHost);
    double avg = 0; for (int i = 0; i < 256 * wps; ++i) avg += (double)h[i];
    printf("  %d wave/SIMD: %.2f cyc/instr/SIMD", wps, avg / (256 * wps) / (iters * 8.0) / wps);
  }
  printf("\n");
}

int main() {
  float* out; unsigned long long* cyc;
  hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
  run<1>("v_fma_f32", out, cyc); run<0>("v_exp_f32", out, cyc); run<2>("v_exp_f16", out, cyc); run<6>("v_rcp_f32", out, cyc);
  run<3>("v_max3_f32", out, cyc); run<4>("v_cvt_pk_f16_f32", out, cyc); run<5>("v_pk_mul_f32", out, cyc);
  return 0;
}

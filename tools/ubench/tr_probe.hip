// Probe: which LDS element lands in which (lane, slot) of ds_read_b64_tr_b16.  LDS holds half(index); lane l reads at
// byte address 8 * l (elements 4l .. 4l+3) and the four returned halves are printed per lane.
// hipcc --offload-arch=gfx950 -O3 tools/ubench/tr_probe.hip -o tools/ubench/tr_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));

__global__ void k(float* out) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[256];
  for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (_Float16)(float)i;
  __syncthreads();
  const unsigned addr = (unsigned)(size_t)lds + threadIdx.x * 8;
  half4_t v;
  asm volatile("ds_read_b64_tr_b16 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
  for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (float)v[j];
}

int main() {
  float* d; hipMalloc(&d, 256 * 4);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
  float h[256]; hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
  for (int l = 0; l < 64; ++l) printf("lane %2d: %3.0f %3.0f %3.0f %3.0f\n", l, h[4 * l], h[4 * l + 1], h[4 * l + 2], h[4 * l + 3]);
  return 0;
}

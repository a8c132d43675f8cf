// Microbenchmark: peak LDS-DMA (buffer_load ... lds, 16 B/lane) throughput per CU from an L2-resident buffer, as a
// function of the DMA instructions each wave keeps in flight (U) and of the waves per CU.  Build + run:
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/dma_bw.hip -o gpurun_out/dma_bw && gpurun_out/dma_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int U>
__global__ __launch_bounds__(256) void dma_kernel(const char* src, unsigned bytes, int iters, int row_bytes,
                                                  unsigned long long* sink) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
  // each DMA instruction: 64 lanes x 16 B; lanes cover (64*16/row_bytes) rows of row_bytes contiguous bytes, rows
  // 4 KB apart (like a K-slice of a row-major operand)
  const int lanes_per_row = row_bytes / 16;
  const unsigned lane_off = (unsigned)(lane / lanes_per_row) * 4096u + (unsigned)(lane % lanes_per_row) * 16u;
  unsigned base = ((blockIdx.x * 4 + wave) * 65536u) % (bytes - 262144u);
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < U; ++u)
      __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (lds_ptr_t)(smem + (wave * U + u) * 1024), 16,
                                               base + lane_off + (unsigned)u * row_bytes, 0, 0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    base = (base + 131072u + (unsigned)U * row_bytes) % (bytes - 262144u);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) sink[blockIdx.x] = t1 - t0;
}

// same access stream with plain buffer loads into VGPRs (no LDS): what the texture path delivers without the DMA
template <int U>
__global__ __launch_bounds__(256) void vgpr_kernel(const char* src, unsigned bytes, int iters, int row_bytes,
                                                   unsigned long long* sink) {
  typedef float f4 __attribute__((ext_vector_type(4)));
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const __amdgpu_buffer_rsrc_t r = __builtin_amdgcn_make_buffer_rsrc((void*)src, 0, bytes, 0x00020000);
  const int lanes_per_row = row_bytes / 16;
  const unsigned lane_off = (unsigned)(lane / lanes_per_row) * 4096u + (unsigned)(lane % lanes_per_row) * 16u;
  unsigned base = ((blockIdx.x * 4 + wave) * 65536u) % (bytes - 262144u);
  f4 acc = {0, 0, 0, 0};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    f4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
      v[u] = __builtin_bit_cast(f4, __builtin_amdgcn_raw_buffer_load_b128(r, base + lane_off + (unsigned)u * row_bytes, 0, 0));
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
    base = (base + 131072u + (unsigned)U * row_bytes) % (bytes - 262144u);
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  if (threadIdx.x == 0) sink[blockIdx.x] = t1 - t0;
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.f) sink[0] = 1;
}

template <int U>
void run(const char* src, unsigned bytes, int blocks_per_cu, int row_bytes, unsigned long long* sink) {
  const int iters = 40000 / U + 50;
  const int nb = 256 * blocks_per_cu;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t lds = (size_t)4 * U * 1024;
  hipLaunchKernelGGL(dma_kernel<U>, dim3(nb), dim3(256), lds, 0, src, bytes, 10, row_bytes, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(dma_kernel<U>, dim3(nb), dim3(256), lds, 0, src, bytes, iters, row_bytes, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[4096];
  hipMemcpy(h, sink, nb * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  double cyc = 0;
  for (int i = 0; i < nb; ++i) cyc += (double)h[i];
  cyc /= nb;
  const double total = (double)nb * 4 * U * 1024.0 * iters;
  {
    hipLaunchKernelGGL(vgpr_kernel<U>, dim3(nb), dim3(256), 0, 0, src, bytes, 10, row_bytes, sink);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(vgpr_kernel<U>, dim3(nb), dim3(256), 0, 0, src, bytes, iters, row_bytes, sink);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms2;
    hipEventElapsedTime(&ms2, e0, e1);
    printf("[vgpr loads %6.2f TB/s] ", total / ms2 / 1e9);
  }
  printf("row %4d B  U=%2d  blocks/CU=%d (%2d waves/CU, %3d KB in flight/CU): %7.2f TB/s  %6.1f B/clk/CU (s_memtime)  "
         "%6.1f B/ns/CU\n", row_bytes, U, blocks_per_cu, 4 * blocks_per_cu, 4 * U * blocks_per_cu, total / ms / 1e9,
         (double)blocks_per_cu * 4 * U * 1024.0 * iters / cyc, total / 256 / (ms * 1e6));
}

int main(int argc, char** argv) {
  const unsigned mb = argc > 1 ? atoi(argv[1]) : 16;        // footprint: 16 MB = 2 MB per XCD slice -> L2 resident
  const unsigned bytes = mb << 20;
  char* src;
  unsigned long long* sink;
  hipMalloc(&src, bytes);
  hipMemset(src, 1, bytes);
  hipMalloc(&sink, 4096 * sizeof(unsigned long long));
  printf("footprint %u MB\n", mb);
  for (int rb : {128, 1024}) {
    for (int bpc : {1, 2, 4}) {
      run<2>(src, bytes, bpc, rb, sink);
      run<4>(src, bytes, bpc, rb, sink);
      run<9>(src, bytes, bpc, rb, sink);
      if (bpc * 4 * 18 <= 150) run<18>(src, bytes, bpc, rb, sink);
    }
  }
  return 0;
}

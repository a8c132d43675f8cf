// Microbenchmark: what the chip SUSTAINS (wall TFLOP/s and effective clock) on dense fp16 MFMA streams with random
// operands, as a function of the MFMA shape, of how accumulators are revisited and of the LDS traffic beside the stream.
// Round 5: the hand-placed conv loop (gemm9.hip) raised the matrix pipe's busy fraction from 0.59 to 0.73 of the cycles
// and the chip answered with 1.57 GHz instead of 2.03 - this probe prices the terms of that trade one by one.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_power.hip -o gpurun_out/mfma_power && gpurun_out/mfma_power
// Every kernel: 256 workgroups x 512 threads (two waves per SIMD), `iters` rounds of 40 (16x16x32) or 20 (32x32x16) MFMA
// slots of equal flops per wave, operands re-read from an LDS image of random fp16 data (or held) as the mode says.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef _Float16 half_t;
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));

// MODE 0: 16x16x32, wave tile 64 x 160 (4 x 10 accumulators), fragments re-read from LDS every round (14 reads / 40 MFMAs)
// MODE 1: 32x32x16, wave tile 64 x 160 (2 x 5), two k16 steps per round, fragments re-read (14 reads / 20 MFMAs)
// MODE 2: as 1, fragments read ONCE (operands constant: no LDS traffic, no operand toggling)
// MODE 3: as 1, the reads are issued but the MFMAs use the held set (LDS traffic, no operand toggling)
// MODE 4: as 1 with the two k16 steps of an accumulator back to back (accumulator chain of two)
// MODE 5: 32x32x16, wave tile 128 x 160 emulated: 4 x 5 MFMAs per k16 step on 2 x 5 accumulators twice (9 reads / 20 MFMAs)
// MODE 6: as 0 without re-reading (16x16x32, constant operands)
template <int MODE>
__global__ __launch_bounds__(512, 2) void k(const half_t* __restrict__ src, float* out, unsigned long long* cyc, int iters) {
  __shared__ __attribute__((aligned(16))) half_t lds[64 * 1024];      // 128 KB image of random halves
  const int tid = threadIdx.x, lane = tid & 63;
  for (int i = tid; i < 64 * 1024 / 8; i += 512) *(half8_t*)&lds[i * 8] = *(const half8_t*)&src[(size_t)i * 8];
  __syncthreads();
  const int wave = tid >> 6;
  const half_t* base = lds + wave * 4096 + lane * 8;      // conflict-free: 64 lanes x 16 B contiguous
  float acc_sum = 0.f;
  unsigned long long t0 = __builtin_readcyclecounter();
  if constexpr (MODE == 0 || MODE == 6) {
    float4_t acc[4][10];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 10; ++j) acc[i][j] = float4_t{0, 0, 0, 0};
    half8_t xf[4], wf[10];
    for (int i = 0; i < 4; ++i) xf[i] = *(const half8_t*)(base + i * 512);
    for (int j = 0; j < 10; ++j) wf[j] = *(const half8_t*)(base + 2048 + j * 512);
    for (int it = 0; it < iters; ++it) {
      const int o = (it & 7) * 8192;
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 10; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
      if (MODE == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[i] = *(const half8_t*)(base + o + i * 512);
#pragma unroll
        for (int j = 0; j < 10; ++j) wf[j] = *(const half8_t*)(base + o + 2048 + j * 512);
      }
    }
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 10; ++j) acc_sum += acc[i][j][0] + acc[i][j][3];
  } else {
    float16_t acc[2][5];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 5; ++j) for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
    half8_t xf[2][4], wf[2][5];      // [k16 step][fragment]
    for (int s = 0; s < 2; ++s) {
      for (int i = 0; i < 4; ++i) xf[s][i] = *(const half8_t*)(base + s * 1024 + i * 512);
      for (int j = 0; j < 5; ++j) wf[s][j] = *(const half8_t*)(base + 4096 + s * 1024 + j * 512);
    }
    half8_t sink[14];
    for (int it = 0; it < iters; ++it) {
      const int o = (it & 7) * 8192;
      if constexpr (MODE == 4) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 5; ++j)
#pragma unroll
            for (int s = 0; s < 2; ++s) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s][j], xf[s][i], acc[i][j], 0, 0, 0);
      } else if constexpr (MODE == 5) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 5; ++j) acc[i & 1][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[0][j], xf[0][i], acc[i & 1][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(wf[s][j], xf[s][i], acc[i][j], 0, 0, 0);
      }
      if constexpr (MODE == 1 || MODE == 4) {
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
          for (int i = 0; i < 2; ++i) xf[s][i] = *(const half8_t*)(base + o + s * 1024 + i * 512);
#pragma unroll
          for (int j = 0; j < 5; ++j) wf[s][j] = *(const half8_t*)(base + o + 4096 + s * 1024 + j * 512);
        }
      } else if constexpr (MODE == 5) {
#pragma unroll
        for (int i = 0; i < 4; ++i) xf[0][i] = *(const half8_t*)(base + o + i * 512);
#pragma unroll
        for (int j = 0; j < 5; ++j) wf[0][j] = *(const half8_t*)(base + o + 4096 + j * 512);
      } else if constexpr (MODE == 3) {
#pragma unroll
        for (int r = 0; r < 14; ++r) {
          sink[r] = *(const half8_t*)(base + o + r * 512);
          asm volatile("" ::"v"(sink[r]));
        }
      }
    }
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 5; ++j) acc_sum += acc[i][j][0] + acc[i][j][7];
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  out[blockIdx.x * 512 + tid] = acc_sum;
  if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MODE>
void run(const char* name, const half_t* src, float* out, unsigned long long* cyc, int iters) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  unsigned long long h[256];
  double clk = 0;
  for (int rep = 0; rep < 4; ++rep) {
    hipEventRecord(e0, 0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(512), 0, 0, src, out, cyc, iters);
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) {
      best = ms;
      hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
      double avg = 0;
      for (int i = 0; i < 256; ++i) avg += (double)h[i];
      clk = avg / 256 / (ms * 1e-3) * 1e-9;      // (s_memtime runs at a fixed 100 MHz on some parts: then this reads 0.1)
    }
  }
  const double flops = 256.0 * 8 * iters * 40 * 2.0 * 16 * 16 * 32;      // every mode: 40 x 16x16x32-equivalents per wave and round
  printf("%-58s %8.3f ms  %7.1f TFLOP/s  s_memtime/wall %.3f GHz\n", name, best, flops / (best * 1e-3) * 1e-12, clk);
}

int main() {
  const size_t n = 64 * 1024;
  std::vector<half_t> h(n);
  srand(7);
  for (size_t i = 0; i < n; ++i) h[i] = (half_t)((rand() / (float)RAND_MAX) * 2.f - 1.f);
  half_t* src; float* out; unsigned long long* cyc;
  hipMalloc(&src, n * 2); hipMalloc(&out, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
  hipMemcpy(src, h.data(), n * 2, hipMemcpyHostToDevice);
  const int iters = 6000;
  for (int pass = 0; pass < 2; ++pass) {
    run<0>("0: 16x16x32, 4x10 acc, operands re-read from LDS", src, out, cyc, iters);
    run<6>("6: 16x16x32, 4x10 acc, operands held", src, out, cyc, iters);
    run<1>("1: 32x32x16, 2x5 acc, operands re-read from LDS", src, out, cyc, iters);
    run<2>("2: 32x32x16, 2x5 acc, operands held", src, out, cyc, iters);
    run<3>("3: 32x32x16, operands held, LDS reads issued beside", src, out, cyc, iters);
    run<4>("4: 32x32x16, re-read, accumulator chains of two", src, out, cyc, iters);
    run<5>("5: 32x32x16, 4x5 MFMAs per 9 reads (128 x 160 wave tile)", src, out, cyc, iters);
  }
  return 0;
}

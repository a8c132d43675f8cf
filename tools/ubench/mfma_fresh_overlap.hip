// Reproducer attempt for the MFMA hazard round 3 ran into in xattn.hip (EXPERIMENTS.md "fused cross-attention sub-block"): an
// accumulator-STARTING v_mfma_f32_16x16x32_f16 (SrcC = inline 0) whose destination is allocated on top of its own A operand,
// followed closely by the dependent next k-step of the same accumulator, lost the second step's contribution in part of the
// destination registers for a timing-dependent share of the waves.  "The exact trigger is not isolated": this file isolates the
// ingredients in hand-written asm (the compiler never sees inside the asm block, so the sequence is exactly what is written):
//
//   PRE   what writes the A registers of MFMA #1 right before it
//           0  nothing (loaded long before)      1  VALU writes (v_mov) of all four A registers + the VALU -> MFMA wait states
//           2  ds_read_b128 into the A registers + s_waitcnt lgkmcnt(0)
//   GAP   what sits between MFMA #1 (v[d:d+3] = A1 x B1 + 0, d == registers of A1) and the dependent MFMA #2 (+= A2 x B2)
//           0..8 independent VALU instructions (v_mov on an unrelated register), or 100 + k: k independent MFMAs on another
//           accumulator
//   POST  what follows MFMA #2
//           0  nothing     1  ds_read_b128 that overwrites the A registers of MFMA #2 right behind it (+ wait)
//
// Every (PRE, GAP, POST) cell runs on every wave of a full chip (2 048 waves x ITER iterations, operands random per lane and
// iteration) and is compared bit for bit against the same two products computed into a destination that overlaps nothing, from
// an accumulator that starts in a zeroed register.  Prints mismatching waves / elements per cell.
//   hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_fresh_overlap.hip -o tools/ubench/mfma_fresh_overlap && tools/ubench/mfma_fresh_overlap
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));

#define GAP_VALU(n) if (GAP == n) asm volatile(REPEAT_##n("v_mov_b32 %0, %0\n\t") : "+v"(dummy));
#define REPEAT_1(x) x
#define REPEAT_2(x) x x
#define REPEAT_3(x) x x x
#define REPEAT_4(x) x x x x
#define REPEAT_6(x) x x x x x x
#define REPEAT_8(x) x x x x x x x x

template <int PRE, int GAP, int POST>
__global__ __launch_bounds__(256) void probe(const half8_t* __restrict__ in, float4_t* __restrict__ out, float4_t* __restrict__ ref,
                                             int iters) {
  __shared__ __attribute__((aligned(16))) half8_t lds[256 * 2];
  const int tid = threadIdx.x, gid = blockIdx.x * 256 + tid;
  for (int it = 0; it < iters; ++it) {
    const half8_t* p = in + ((size_t)(it * gridDim.x + blockIdx.x) * 256 + tid) * 4;
    half8_t a1 = p[0], b1 = p[1], a2 = p[2], b2 = p[3];
    lds[tid] = a1;
    lds[256 + tid] = a2;
    __syncthreads();
    // reference: nothing overlaps, the accumulator starts in a zeroed register
    float4_t z = {0.f, 0.f, 0.f, 0.f};
    asm volatile("" : "+v"(z));
    float4_t r = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, z, 0, 0, 0);
    asm volatile("" ::"v"(a1), "v"(b1));
    r = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b2, r, 0, 0, 0);
    // the pattern: d = a1's own registers
    float4_t d = __builtin_bit_cast(float4_t, a1);
    float4_t a2r = __builtin_bit_cast(float4_t, a2);
    float dummy = (float)tid;
    float4_t acc2 = {0.f, 0.f, 0.f, 0.f};
    const unsigned la1 = (unsigned)(tid * 16), la2 = (unsigned)((256 + tid) * 16);
    asm volatile("s_nop 7" : "+v"(d), "+v"(a2r), "+v"(b1), "+v"(b2));      // (nothing the compiler wrote is still in flight)
    if (PRE == 1)
      asm volatile("v_mov_b32 %0, %0\n\tv_mov_b32 %1, %1\n\tv_mov_b32 %2, %2\n\tv_mov_b32 %3, %3\n\ts_nop 1"      // VALU writes of all four A registers
                   : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));
    if (PRE == 2) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(d) : "v"(la1) : "memory");
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %0, %1, 0" : "+v"(d) : "v"(b1));
    GAP_VALU(1) GAP_VALU(2) GAP_VALU(3) GAP_VALU(4) GAP_VALU(6) GAP_VALU(8)
    if (GAP >= 100) {
      for (int k = 0; k < GAP - 100; ++k) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc2) : "v"(a2r), "v"(b2));
    }
    asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(d) : "v"(a2r), "v"(b2));
    if (POST == 1) asm volatile("ds_read_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(a2r) : "v"(la2) : "memory");
    asm volatile("s_nop 7\n\ts_nop 7" : "+v"(d), "+v"(a2r), "+v"(acc2), "+v"(dummy));
    out[(size_t)it * gridDim.x * 256 + gid] = d;
    ref[(size_t)it * gridDim.x * 256 + gid] = r;
    __syncthreads();
  }
}

static half8_t* g_in;
static float4_t *g_out, *g_ref, *h_out, *h_ref;
constexpr int BLOCKS = 512, ITER = 64;

template <int PRE, int GAP, int POST>
void run() {
  const size_t n = (size_t)BLOCKS * 256 * ITER;
  hipMemset(g_out, 0xff, n * sizeof(float4_t));
  hipLaunchKernelGGL((probe<PRE, GAP, POST>), dim3(BLOCKS), dim3(256), 0, 0, g_in, g_out, g_ref, ITER);
  hipDeviceSynchronize();
  hipMemcpy(h_out, g_out, n * sizeof(float4_t), hipMemcpyDeviceToHost);
  hipMemcpy(h_ref, g_ref, n * sizeof(float4_t), hipMemcpyDeviceToHost);
  size_t bad = 0, badw = 0;
  for (size_t w = 0; w < n / 64; ++w) {
    size_t b = 0;
    for (int l = 0; l < 64; ++l)
      for (int e = 0; e < 4; ++e) b += ((const unsigned*)&h_out[w * 64 + l])[e] != ((const unsigned*)&h_ref[w * 64 + l])[e];
    bad += b; badw += b != 0;
  }
  printf("PRE %d  GAP %3d  POST %d : %8zu of %zu elements differ, %6zu of %zu wave-iterations\n", PRE, GAP, POST, bad, n * 4, badw, n / 64);
}

int main() {
  const size_t n = (size_t)BLOCKS * 256 * ITER;
  hipMalloc(&g_in, n * 4 * sizeof(half8_t));
  hipMalloc(&g_out, n * sizeof(float4_t));
  hipMalloc(&g_ref, n * sizeof(float4_t));
  h_out = (float4_t*)malloc(n * sizeof(float4_t));
  h_ref = (float4_t*)malloc(n * sizeof(float4_t));
  half8_t* h = (half8_t*)malloc(n * 4 * sizeof(half8_t));
  srand(1);
  for (size_t i = 0; i < n * 4; ++i)
    for (int j = 0; j < 8; ++j) h[i][j] = (_Float16)((rand() % 2001 - 1000) * 1e-3f);
  hipMemcpy(g_in, h, n * 4 * sizeof(half8_t), hipMemcpyHostToDevice);
#define ROW(PRE, POST) run<PRE, 0, POST>(); run<PRE, 1, POST>(); run<PRE, 2, POST>(); run<PRE, 3, POST>(); run<PRE, 4, POST>(); \
  run<PRE, 6, POST>(); run<PRE, 8, POST>(); run<PRE, 101, POST>(); run<PRE, 102, POST>(); run<PRE, 104, POST>();
  ROW(0, 0) ROW(1, 0) ROW(2, 0) ROW(0, 1) ROW(2, 1)
  return 0;
}

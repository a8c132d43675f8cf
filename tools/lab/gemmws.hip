// Weight-stationary streaming GEMM for the HBM-bound N = K = 320 layers of the 64 x 64 level (to_out / to_q / proj_in /
// proj_out of SD1.5's first and last levels, the 96 x 96 level of SD2.1):
//
//   C[m][n] = alpha * (sum_k A[m][k] * B[n][k] + bias[n]) + residual[m][n],   M large, N = K = 320
//
// These launches move 84-126 MB for 13 GFLOP: the tile kernel (gemm2.hip, 128 x 160 tiles, K loop of five steps) spends its
// time in exposed latencies - every workgroup waits for its first operand tile, walks five K steps in lock step with the
// workgroup it shares the CU with, then waits again for the residual and for its stores - and lands at 40 us where the bytes
// take 22.  Here nothing but the activation rows, the residual and the output ever moves more than once:
//   * one persistent 4-wave workgroup per CU; wave w keeps the weights of output columns [80 w, 80 w + 80) in REGISTERS for the
//     whole launch (5 column tiles x 10 k-steps x 4 VGPRs = 200 VGPRs; one wave per SIMD has 512),
//   * row tiles of 64 x 320 activations stream through a double-buffered LDS image by LDS-DMA (the swizzled [rows][64]
//     k-block layout of gemm2.hip); the DMA of tile t + 1 is issued before tile t is computed, the residual of tile t + 1
//     after tile t's stores, and both are waited for with COUNTED vmcnt (loads and stores retire in issue order), so a
//     workgroup never drains its memory queue,
//   * 20 MFMAs per fragment read (weights need none), fp32 accumulators through a 32-row LDS staging slab, whole 640-byte
//     output rows stored as 16-byte pieces.
// Same MFMA sequence and the same fp32 epilogue arithmetic as gemm2.hip's unsplit path: bit-identical results.
#include "gemm_params.h"
#include <stdlib.h>

namespace {

constexpr int WN = 320, WK = 320, WBM = 64, WNTHR = 256;
constexpr int KB = WK / 64;                       // 64-deep k-blocks of the LDS image
constexpr int ABUF = KB * WBM * 64;               // halves per activation buffer (40 KB)
constexpr int SROWS = 32, OPF = WN + 4;           // staging slab: 32 rows x 324 floats
constexpr unsigned OOB = 0x80000000u;
constexpr int LDS_BYTES = 2 * ABUF * 2 + SROWS * OPF * 4;

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef unsigned u4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <bool RES>
__global__ __launch_bounds__(WNTHR) void gemmws_kernel(const GemmParams p, int ntiles, unsigned a_bytes, unsigned r_bytes,
                                                        unsigned c_bytes) {
  __shared__ __attribute__((aligned(16))) unsigned char smem_raw[LDS_BYTES];
  half_t* const abuf = reinterpret_cast<half_t*>(smem_raw);
  float* const stg = reinterpret_cast<float*>(smem_raw + 2 * ABUF * 2);
  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, l16 = lane & 15;
  const __amdgpu_buffer_rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc((void*)p.A, 0, a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rR = __builtin_amdgcn_make_buffer_rsrc((void*)(RES ? p.res : p.A), 0, RES ? r_bytes : 0u, 0x00020000);
  const __amdgpu_buffer_rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(p.C, 0, c_bytes, 0x00020000);

  // ---- the weights of this wave's 80 output columns, and their bias, for the whole launch -----------------------------
  half8_t wreg[5][10];
#pragma unroll
  for (int j = 0; j < 5; ++j)
#pragma unroll
    for (int ks = 0; ks < 10; ++ks)
      wreg[j][ks] = ld_half8(p.B + (size_t)(80 * wave + 16 * j + l16) * p.ldb + 32 * ks + 8 * g);
  // (through a descriptor whose range is empty without a bias: unconditional loads - a per-load "if (p.bias)" makes hipcc
  // branch around each one and wait for it, five serial round trips before the first tile)
  const __amdgpu_buffer_rsrc_t rBias = __builtin_amdgcn_make_buffer_rsrc((void*)(p.bias ? p.bias : p.A), 0, p.bias ? WN * 2u : 0u, 0x00020000);
  typedef unsigned u2_t __attribute__((ext_vector_type(2)));
  u2_t braw[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) braw[j] = __builtin_amdgcn_raw_buffer_load_b64(rBias, (unsigned)(80 * wave + 16 * j + 4 * g) * 2u, 0, 0);

  // ---- LDS-DMA of one 64 x 320 activation tile: wave w fills row groups w and w + 4 (8 rows each) of the 5 k-blocks ------
  const int lr = lane >> 3, lq = lane & 7;
  auto dma_tile = [&](int tile, int buf) {
    unsigned v[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      const int r = 8 * (wave + 4 * s) + lr;
      const long m = (long)tile * WBM + r;
      v[s] = (tile < ntiles && m < p.M) ? (unsigned)m * (unsigned)p.lda * 2u + (unsigned)(lq ^ ((r >> 1) & 7)) * 16u : OOB;
    }
#pragma unroll
    for (int i = 0; i < 2 * KB; ++i) {
      const int kb = i >> 1, s = i & 1;
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rA, (lds_ptr_t)(abuf + buf * ABUF + kb * (WBM * 64) + (wave + 4 * s) * (8 * 64)), 16,
                                               v[s], (unsigned)kb * 128u, 0, 0);
    }
  };
  // phase-2 map of the epilogue: thread -> row er of the 32-row slab, 16-byte pieces pc + 8 k of its 40
  const int er = tid >> 3, pc = tid & 7;
  half8_t rv[2][5];
  auto load_res = [&](int tile) {
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const long m = (long)tile * WBM + SROWS * h + er;
      const unsigned vo = (tile < ntiles && m < p.M) ? (unsigned)(((size_t)m * p.ldr + 8 * pc) * 2) : OOB;
#pragma unroll
      for (int k = 0; k < 5; ++k) rv[h][k] = __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(rR, vo, k * 128, 0));
    }
  };

  // fragment read addresses (halves inside a buffer) of this lane: row 16 i + l16, k-step parity ks2
  int x_ad[4][2];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int row = 16 * i + l16;
#pragma unroll
    for (int ks2 = 0; ks2 < 2; ++ks2) x_ad[i][ks2] = row * 64 + (((ks2 * 4 + g) ^ ((row >> 1) & 7)) << 3);
  }

  // Waits are COUNTED, and only loads are ever counted on: loads retire in issue order among themselves, stores among
  // themselves, but a store may retire before an older load (a first version that counted across this tile's stores read
  // LDS tiles that had not landed - only when a second stream shared the chip).  The one wait of an iteration sits BEFORE the
  // tile's stores are issued; the stores of the previous tile, a whole compute phase old, are the only ones that can still be
  // pending there, and if they are the wait is merely longer than needed.
  int t = blockIdx.x;
  const int G = gridDim.x;
  dma_tile(t, 0);
  dma_tile(t + G, 1);
  if (RES) load_res(t);
  if (RES) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");      // tile t has landed (younger: tile t + G, the residual)
  else asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
  lds_barrier();
  float4_t b4[5];
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const half4_t b = __builtin_bit_cast(half4_t, braw[j]);
    b4[j] = float4_t{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
  }
  int buf = 0;
  for (; t < ntiles; t += G, buf ^= 1) {
    float4_t acc[4][5];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 5; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
    const half_t* const sb = abuf + buf * ABUF;
#pragma unroll
    for (int ks = 0; ks < 10; ++ks) {
      half8_t xf[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) xf[i] = ld_half8(sb + (ks >> 1) * (WBM * 64) + x_ad[i][ks & 1]);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 5; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wreg[j][ks], xf[i], acc[i][j], 0, 0, 0);
    }
    lds_barrier();                               // every wave is done with this buffer (and with the previous tile's slab)
    dma_tile(t + 2 * G, buf);                    // ... which takes the tile after next (out of range: zero fill, same count)
    // tile t + G and this tile's residual have landed when at most the 10 DMA instructions just issued are outstanding
    asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    // ---- epilogue: two 32-row slabs through LDS; lane holds C[m = 16 i + l16][n = 80 w + 16 j + 4 g .. + 3] --------------
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if (h) lds_barrier();                      // slab 0 has been read
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int j = 0; j < 5; ++j)
          *reinterpret_cast<float4_t*>(&stg[(16 * ii + l16) * OPF + 80 * wave + 16 * j + 4 * g]) = acc[2 * h + ii][j] + b4[j];
      lds_barrier();                             // (h = 0: also publishes tile t + G - every wave has passed its wait above)
      const long m = (long)t * WBM + SROWS * h + er;
      const unsigned vo = m < p.M ? (unsigned)(((size_t)m * p.ldc + 8 * pc) * 2) : OOB;
#pragma unroll
      for (int k = 0; k < 5; ++k) {
        const float* const s = stg + er * OPF + 8 * (pc + 8 * k);
        const float4_t v0 = *reinterpret_cast<const float4_t*>(s), v1 = *reinterpret_cast<const float4_t*>(s + 4);
        float v[8] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
        half8_t o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          v[e] = v[e] * p.alpha + (RES ? (float)rv[h][k][e] : 0.f);
          o[e] = (half_t)v[e];
        }
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u4_t, o), rC, vo, k * 128, 0);
      }
    }
    if (RES) load_res(t + G);
  }
}

bool ws_enabled() {
  const char* e = getenv("SKG_GEMMWS");      // lab build: opt-in, read per launch (tools/lab/gemmws_bench.py flips it)
  return e && atoi(e) != 0;
}

}  // namespace

bool skg_gemmws_eligible(const GemmParams& p, int mode) {
  if (!ws_enabled() || mode != MODE_DIRECT || p.N != WN || p.K != WK || p.M < 32768) return false;
  if (p.flags || p.gn_partial || p.aux || p.c_lo || p.res_lo || p.ntaps || p.up2 || p.K2) return false;
  if (p.lda % 8 || p.ldb % 8 || p.ldc % 8 || (p.res && p.ldr % 8)) return false;
  if ((reinterpret_cast<uintptr_t>(p.A) | reinterpret_cast<uintptr_t>(p.B) | reinterpret_cast<uintptr_t>(p.C) |
       reinterpret_cast<uintptr_t>(p.res)) & 15)
    return false;
  if (p.bias && (reinterpret_cast<uintptr_t>(p.bias) & 7)) return false;
  const unsigned long long a = ((unsigned long long)(p.M - 1) * p.lda + p.K) * 2ull, c = ((unsigned long long)(p.M - 1) * p.ldc + p.N) * 2ull;
  const unsigned long long r = p.res ? ((unsigned long long)(p.M - 1) * p.ldr + p.N) * 2ull : 0;
  return a < 0x7fffffffull && c < 0x7fffffffull && r < 0x7fffffffull;
}

bool skg_gemmws_try_launch(const GemmParams& p, int mode, hipStream_t st) {
  if (!skg_gemmws_eligible(p, mode)) return false;
  const int ntiles = skg_cdiv(p.M, WBM);
  const unsigned a = (unsigned)(((size_t)(p.M - 1) * p.lda + p.K) * 2), c = (unsigned)(((size_t)(p.M - 1) * p.ldc + p.N) * 2);
  const unsigned r = p.res ? (unsigned)(((size_t)(p.M - 1) * p.ldr + p.N) * 2) : 0u;
  const int grid = ntiles < 256 ? ntiles : 256;
  if (p.res) hipLaunchKernelGGL(gemmws_kernel<true>, dim3(grid), dim3(WNTHR), 0, st, p, ntiles, a, r, c);
  else hipLaunchKernelGGL(gemmws_kernel<false>, dim3(grid), dim3(WNTHR), 0, st, p, ntiles, a, r, c);
  return true;
}

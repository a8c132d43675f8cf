// Library-level entry points of libskg.so: ABI version and last-error text.
#include "common.h"
#include <stdio.h>

static thread_local char g_err[256] = "";

void skg_set_error(const char* what, hipError_t e) {
  snprintf(g_err, sizeof(g_err), "%s: %s", what, hipGetErrorString(e));
}

extern "C" int skg_abi_version(void) { return SKG_ABI_VERSION; }
extern "C" const char* skg_last_error(void) { return g_err; }

// ---------------------------------------------------------------------------------------------- box calibration probe
// What THIS box sustains on a bare fp16 MFMA stream (bench.py's `config.box_mfma_tflops`): 256 workgroups x 8 waves (two per
// SIMD), every wave a 64 x 160 accumulator tile (4 x 10 tiles of 16 x 16 x 32), operands = pseudo-random halves in [-1, 1)
// held in registers, `iters` rounds of 40 MFMAs.  No LDS, no memory traffic inside the loop: the figure moves with the
// sustained shader clock under the package power limit and with nothing else (round 5: the same code read 6.4 - 6.9 images/s
// on the boxes drawn; `tools/ubench/mfma_power.hip` mode 6 is the stand-alone form of this kernel).
namespace {
__device__ __forceinline__ half_t probe_half(unsigned x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return (half_t)(((int)(x & 0xffff) - 32768) * (1.f / 32768.f));
}
__global__ __launch_bounds__(512, 2) void box_probe_kernel(float* __restrict__ out, int iters) {
  const unsigned tid = blockIdx.x * 512 + threadIdx.x;
  half8_t xf[4], wf[10];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int e = 0; e < 8; ++e) xf[i][e] = probe_half(tid * 131u + i * 8 + e);
#pragma unroll
  for (int j = 0; j < 10; ++j)
#pragma unroll
    for (int e = 0; e < 8; ++e) wf[j][e] = probe_half(tid * 257u + 4096 + j * 8 + e);
  float4_t acc[4][10];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 10; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 10; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
  }
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 10; ++j) s += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
  out[tid] = s;
}
}  // namespace

extern "C" int skg_box_probe_mfma(float* out, int iters, void* stream) {
  SKG_REQUIRE(out != nullptr && iters > 0 && iters <= (1 << 24));
  hipLaunchKernelGGL(box_probe_kernel, dim3(SKG_BOX_PROBE_WORKGROUPS), dim3(512), 0, (hipStream_t)stream, out, iters);
  SKG_CHECK_LAUNCH("skg_box_probe_mfma");
  return SKG_OK;
}

// Winograd F(2x2, 3x3) for the stride-1 3x3 convolutions of the small maps (round 6; include/skg.h skg_conv3x3_wino_f16).
//
// The 16 x 16 / 8 x 8 levels of the UNet launch at most one 128 x 160 tile per CU: what bounds them is operand bytes per
// flop and per-launch fixed cost, not the matrix pipe.  F(2,3) computes a 2 x 2 output tile from a 4 x 4 input tile with 16 products per
// (cin, cout) instead of 36 - 2.25 x fewer MFMA flops - and turns one convolution over M pixels into 16 INDEPENDENT GEMMs over M / 4 tile
// positions, i.e. 4 x more workgroups of the same 128 x 160 tile on exactly the launches that under-fill the chip:
//
//     V = B^T d B   (input transform, this file)      [Mt][16][Cin]   fp16, one rounding of sums of <= 4 fp16 inputs formed in fp32
//     M_c = V_c . U_c^T,  c = 0..15                   the existing split-K launch of gemm2.hip on V [Mt][16 Cin] x U [Cout][16 Cin] with the
//                                                     K range of slab c = component c: 16 fp32 slabs, no new GEMM code
//     Y = A^T M A   (output transform, this file)     replaces the split-K reduce: +/-1 combinations of 9 of the 16 slabs per output pixel,
//                                                     then the ordinary epilogue (bias, residual, ReLU, pair output)
//     U = G g G^T   (weights, host side at pack time: unet.pack_conv_wino)
//
// Numerics: U and V are a second fp16 rounding of weights and activations (fp32 accumulation, fp32 output transform): priced on the CPU
// oracle before this file existed (tools/eps_winograd.py, profiles/r06_eps_winograd_cpu.txt) - default mode eps rel 1.07e-3 -> 1.09e-3.
#include "gemm_params.h"

namespace {

// one thread: one tile position x 8 channels.  X [rows*IH*IW][ldx] -> V [Mt][16*Cin], component c = 4 i + j at columns [c Cin, (c+1) Cin)
__global__ __launch_bounds__(256) void wino_in_kernel(const half_t* __restrict__ X, int ldx, half_t* __restrict__ V, int rows, int IH, int IW,
                                                      int Cin, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int c8 = Cin >> 3;
  const int ch = (int)(idx % c8) * 8;
  const long mt = idx / c8;
  const int TW = IW >> 1, TH = IH >> 1;
  const int tj = (int)(mt % TW), ti = (int)((mt / TW) % TH);
  const long r = mt / ((long)TW * TH);
  const half_t* base = X + (r * IH * IW) * (long)ldx + ch;
  float t[4][4][8];      // B^T d (rows transformed), per column
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int x = 2 * tj - 1 + b;
    float d[4][8];
#pragma unroll
    for (int a = 0; a < 4; ++a) {
      const int y = 2 * ti - 1 + a;
      half8_t v = zero_half8();
      if (x >= 0 && x < IW && y >= 0 && y < IH) v = ld_half8(base + ((long)y * IW + x) * ldx);
#pragma unroll
      for (int e = 0; e < 8; ++e) d[a][e] = (float)v[e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      t[0][b][e] = d[0][e] - d[2][e];
      t[1][b][e] = d[1][e] + d[2][e];
      t[2][b][e] = d[2][e] - d[1][e];
      t[3][b][e] = d[1][e] - d[3][e];
    }
  }
  half_t* out = V + mt * (16L * Cin) + ch;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    half8_t v0, v1, v2, v3;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
      v0[e] = (half_t)(t[a][0][e] - t[a][2][e]);
      v1[e] = (half_t)(t[a][1][e] + t[a][2][e]);
      v2[e] = (half_t)(t[a][2][e] - t[a][1][e]);
      v3[e] = (half_t)(t[a][1][e] - t[a][3][e]);
    }
    st_half8(out + (long)(4 * a + 0) * Cin, v0);
    st_half8(out + (long)(4 * a + 1) * Cin, v1);
    st_half8(out + (long)(4 * a + 2) * Cin, v2);
    st_half8(out + (long)(4 * a + 3) * Cin, v3);
  }
}

// one thread: one tile position x 4 output columns.  slabs [16][Mt][N] fp32 -> the four pixels of the tile in Y [rows*IH*IW][ldc]
__global__ __launch_bounds__(256) void wino_out_kernel(GemmParams p, const float* __restrict__ slabs, int IH, int IW, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int n4 = p.N >> 2;
  const int n = (int)(idx % n4) * 4;
  const long mt = idx / n4;
  const size_t slab = (size_t)p.M * p.N;      // (p.M = Mt)
  const float* src = slabs + (size_t)mt * p.N + n;
  float4_t s[4][2];      // M A (columns transformed): s[i][0] = m_i0 + m_i1 + m_i2, s[i][1] = m_i1 - m_i2 - m_i3
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float4_t m0 = *reinterpret_cast<const float4_t*>(src + (size_t)(4 * i + 0) * slab);
    const float4_t m1 = *reinterpret_cast<const float4_t*>(src + (size_t)(4 * i + 1) * slab);
    const float4_t m2 = *reinterpret_cast<const float4_t*>(src + (size_t)(4 * i + 2) * slab);
    const float4_t m3 = *reinterpret_cast<const float4_t*>(src + (size_t)(4 * i + 3) * slab);
    s[i][0] = m0 + m1 + m2;
    s[i][1] = m1 - m2 - m3;
  }
  const int TW = IW >> 1, TH = IH >> 1;
  const int tj = (int)(mt % TW), ti = (int)((mt / TW) % TH);
  const long r = mt / ((long)TW * TH);
  float4_t bias = {0.f, 0.f, 0.f, 0.f};
  if (p.bias) {
    const half4_t b = ld_half4(p.bias + n);
    bias = float4_t{(float)b[0], (float)b[1], (float)b[2], (float)b[3]};
  }
#pragma unroll
  for (int a = 0; a < 2; ++a)
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      float4_t v = a == 0 ? s[0][b] + s[1][b] + s[2][b] : s[1][b] - s[2][b] - s[3][b];      // A^T (M A)
      v = (v + bias) * p.alpha;
      const size_t m = (size_t)((r * IH + 2 * ti + a) * IW + 2 * tj + b);
      if (p.res) {
        const half4_t q = ld_half4(p.res + m * p.ldr + n);
        v[0] += (float)q[0]; v[1] += (float)q[1]; v[2] += (float)q[2]; v[3] += (float)q[3];
      }
      if (p.res_lo) {
        const half4_t q = ld_half4(p.res_lo + m * p.ldr + n);
        v[0] += (float)q[0]; v[1] += (float)q[1]; v[2] += (float)q[2]; v[3] += (float)q[3];
      }
      if (p.flags & SKG_EPI_RELU) {
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
      }
      const half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
      st_half4(reinterpret_cast<half_t*>(p.C) + m * p.ldc + n, o);
      if (p.c_lo) {
        const half4_t l = {(half_t)(v[0] - (float)o[0]), (half_t)(v[1] - (float)o[1]), (half_t)(v[2] - (float)o[2]), (half_t)(v[3] - (float)o[3])};
        st_half4(p.c_lo + m * p.ldc + n, l);
      }
    }
}

}  // namespace

void skg_wino_in_launch(const half_t* X, int ldx, half_t* V, int rows, int IH, int IW, int Cin, hipStream_t st) {
  const long total = (long)rows * (IH / 2) * (IW / 2) * (Cin / 8);
  hipLaunchKernelGGL(wino_in_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, X, ldx, V, rows, IH, IW, Cin, total);
}

void skg_wino_out_launch(const GemmParams& p, const float* slabs, int IH, int IW, hipStream_t st) {
  const long total = (long)p.M * (p.N / 4);
  hipLaunchKernelGGL(wino_out_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p, slabs, IH, IW, total);
}

// fp16 MFMA GEMM and 3x3 implicit-GEMM convolution for gfx950.
//
//   C[m][n] = epi(alpha * (sum_k A[m][k] * B[n][k] + bias[n]) + residual[m][n])
//
// One kernel template serves the dense GEMM and the four conv gather modes: only the A-operand
// address generator differs (the K axis of a conv is (ky, kx, cin) with cin contiguous, so a
// BK = 32 slice always lies inside one filter tap because Cin % 32 == 0).
//
// Tiling (v1): 128 x BN x 32 block tile, 4 waves as 2(M) x 2(N), each wave 64 x BN/2 out of
// 16x16x32 f16 MFMAs.  Operands are staged global -> VGPR -> LDS (padded pitch, conflict-free
// ds_read_b128) with a register prefetch of the next K slice and ONE barrier per slice.
// The MFMA is issued "swapped" (weights as the A operand, activations as B) so that each lane's
// four accumulator registers are four CONSECUTIVE output channels of one pixel: the epilogue
// then does 8-byte bias/residual loads and 8-byte stores instead of 2-byte ones.
#include "gemm_params.h"
#include <mutex>
#include <vector>

namespace {

constexpr int BM = 128;
constexpr int BK = 32;
constexpr int PITCH = 40;   // halves; 80-byte rows keep ds_read_b128 of 16 rows conflict-free

template <int BN, int MODE>
__global__ __launch_bounds__(256) void gemm_kernel(const GemmParams p) {
  constexpr int WN = BN / 2;
  constexpr int NT = WN / 16;
  constexpr int MT = 4;
  constexpr int BPT = BN / 64;   // B 16-byte pieces per thread
  __shared__ __attribute__((aligned(16))) half_t As[2][BM * PITCH];
  __shared__ __attribute__((aligned(16))) half_t Bs[2][BN * PITCH];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int g = lane >> 4, l16 = lane & 15;
  const int m0 = blockIdx.y * BM;
  const int n0 = blockIdx.x * BN;

  // ---- per-thread loader coordinates ------------------------------------------------------
  const int lrow = tid >> 2;          // 0..63
  const int lkc = (tid & 3) * 8;      // k offset of this thread's 16-byte piece
  // A rows lrow and lrow + 64
  bool a_ok[2];
  const half_t* a_base[2];
  int a_oy[2], a_ox[2];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const int m = m0 + lrow + 64 * q;
    a_ok[q] = m < p.M;
    if (MODE == MODE_DIRECT) {
      a_base[q] = p.A + (size_t)(a_ok[q] ? m : 0) * p.lda + lkc;
      a_oy[q] = a_ox[q] = 0;
    } else {
      const int mm = a_ok[q] ? m : 0;
      const int ohw = p.OH * p.OW;
      const int b = mm / ohw;
      const int r = mm - b * ohw;
      a_oy[q] = r / p.OW;
      a_ox[q] = r - a_oy[q] * p.OW;
      a_base[q] = p.A + (size_t)b * p.IH * p.IW * p.lda + lkc;
    }
  }
  bool b_ok[BPT];
  const half_t* b_base[BPT];
#pragma unroll
  for (int q = 0; q < BPT; ++q) {
    const int n = n0 + lrow + 64 * q;
    b_ok[q] = n < p.N;
    b_base[q] = p.B + (size_t)(b_ok[q] ? n : 0) * p.ldb + lkc;
  }

  half8_t ra[2], rb[BPT];
  auto load_tiles = [&](int kt) {
    const int k0 = kt * BK;
    if (MODE == MODE_DIRECT) {
#pragma unroll
      for (int q = 0; q < 2; ++q) ra[q] = a_ok[q] ? ld_half8(a_base[q] + k0) : zero_half8();
    } else {
      const int tap = k0 / p.Cin;
      const int c0 = k0 - tap * p.Cin;
      const int ky = tap / 3, kx = tap - ky * 3;
#pragma unroll
      for (int q = 0; q < 2; ++q) {
        int ty = a_oy[q] + ky - 1, tx = a_ox[q] + kx - 1;
        bool ok = a_ok[q];
        int iy, ix;
        if (MODE == MODE_S1) {
          iy = ty; ix = tx;
          ok = ok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
        } else if (MODE == MODE_S2 || MODE == MODE_S2A) {
          iy = ty + a_oy[q] + (MODE == MODE_S2A); ix = tx + a_ox[q] + (MODE == MODE_S2A);     // 2*o + k - 1 (+1)
          ok = ok && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
        } else if (MODE == MODE_UP2) {
          ok = ok && ty >= 0 && ty < p.OH && tx >= 0 && tx < p.OW;
          iy = ty >> 1; ix = tx >> 1;
        } else {   // MODE_S2T
          ok = ok && ty >= 0 && tx >= 0 && !(ty & 1) && !(tx & 1);
          iy = ty >> 1; ix = tx >> 1;
          ok = ok && iy < p.IH && ix < p.IW;
        }
        ra[q] = ok ? ld_half8(a_base[q] + ((size_t)iy * p.IW + ix) * p.lda + c0) : zero_half8();
      }
    }
#pragma unroll
    for (int q = 0; q < BPT; ++q) rb[q] = b_ok[q] ? ld_half8(b_base[q] + k0) : zero_half8();
  };
  auto store_tiles = [&](int buf) {
#pragma unroll
    for (int q = 0; q < 2; ++q) st_half8(&As[buf][(lrow + 64 * q) * PITCH + lkc], ra[q]);
#pragma unroll
    for (int q = 0; q < BPT; ++q) st_half8(&Bs[buf][(lrow + 64 * q) * PITCH + lkc], rb[q]);
  };

  float4_t acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};

  const int KT = p.K / BK;
  load_tiles(0);
  store_tiles(0);
  __syncthreads();

  for (int kt = 0; kt < KT; ++kt) {
    const int cur = kt & 1;
    const bool more = kt + 1 < KT;
    if (more) load_tiles(kt + 1);
    half8_t xf[MT], wf[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) xf[i] = ld_half8(&As[cur][(wm * 64 + i * 16 + l16) * PITCH + g * 8]);
#pragma unroll
    for (int j = 0; j < NT; ++j) wf[j] = ld_half8(&Bs[cur][(wn * WN + j * 16 + l16) * PITCH + g * 8]);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[j], xf[i], acc[i][j], 0, 0, 0);
    if (more) store_tiles(cur ^ 1);
    __syncthreads();
  }

  // ---- epilogue: lane holds C[m = .. + l16][n = .. + 4g .. 4g+3] ---------------------------
  const bool relu = p.flags & SKG_EPI_RELU;
  const bool f32out = p.flags & SKG_EPI_OUT_F32;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int m = m0 + wm * 64 + i * 16 + l16;
    if (m >= p.M) continue;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int n = n0 + wn * WN + j * 16 + g * 4;
      if (n >= p.N) continue;
      float4_t v = acc[i][j];
      if (p.bias) {
        const half4_t b = ld_half4(p.bias + n);
        v[0] += (float)b[0]; v[1] += (float)b[1]; v[2] += (float)b[2]; v[3] += (float)b[3];
      }
      v *= p.alpha;
      if (p.res) {
        const half4_t r = ld_half4(p.res + (size_t)m * p.ldr + n);
        v[0] += (float)r[0]; v[1] += (float)r[1]; v[2] += (float)r[2]; v[3] += (float)r[3];
      }
      if (relu) {
        v[0] = fmaxf(v[0], 0.f); v[1] = fmaxf(v[1], 0.f); v[2] = fmaxf(v[2], 0.f); v[3] = fmaxf(v[3], 0.f);
      }
      if (f32out) {
        *reinterpret_cast<float4_t*>(reinterpret_cast<float*>(p.C) + (size_t)m * p.ldc + n) = v;
      } else {
        half4_t o = {(half_t)v[0], (half_t)v[1], (half_t)v[2], (half_t)v[3]};
        st_half4(reinterpret_cast<half_t*>(p.C) + (size_t)m * p.ldc + n, o);
      }
    }
  }
}

// wide (128-column) tiles when N fills them and there are enough blocks to cover the 256 CUs
inline bool use_wide(int M, int N) { return (N % 128 == 0) && ((long)skg_cdiv(M, BM) * (N / 128) >= 256); }

template <int MODE>
int launch_plain(const GemmParams& p, hipStream_t st);

// With p.gn_partial set the launch also leaves the GroupNorm partial sums of its output behind: from the kernel's own
// epilogue when the instantiation that runs can (v8 256 x 320, v2 128 x 160), else from the stand-alone statistics pass.
template <int MODE>
int launch(const GemmParams& p, hipStream_t st) {
  if (!p.gn_partial) return launch_plain<MODE>(p, st);
#ifdef SKG_LAB
  const bool v9 = skg_gemm9_eligible(p, MODE);      // (no statistics epilogue: the stand-alone pass follows)
#else
  constexpr bool v9 = false;
#endif
  const bool fused = v9 ? false : skg_gemm8_eligible(p, MODE) ? skg_gemm8_fuses_gn(p, MODE) : skg_gemm2_fuses_gn(p, MODE);      // (the k-pair kernel declines launches that ask for statistics)
  const int rc = launch_plain<MODE>(p, st);
  if (rc != SKG_OK || fused) return rc;
  skg_gn_partial_launch((const half_t*)p.C, p.ldc, p.M / p.gn_hw, p.gn_hw, p.N, p.gn_groups, p.gn_hw / 128, p.gn_partial, st);
  SKG_CHECK_LAUNCH("skg_gemm (statistics pass)");
  return SKG_OK;
}

template <int MODE>
int launch_plain(const GemmParams& p, hipStream_t st) {
#ifdef SKG_LAB      // withdrawn round-3 experiment (tools/lab/gemmws.hip, EXPERIMENTS.md): lab build only, SKG_GEMMWS=1
  if (skg_gemmws_try_launch(p, MODE, st)) {
    SKG_CHECK_LAUNCH("skg_gemm (ws)");
    return SKG_OK;
  }
#endif
#ifdef SKG_LAB      // round-5 experiment (tools/lab/gemm9.hip, EXPERIMENTS.md "the hand-placed K loop"): lab build only, SKG_GEMM9=<geometry><variant>
  if (skg_gemm9_try_launch(p, MODE, st)) {
    SKG_CHECK_LAUNCH("skg_gemm (v9)");
    return SKG_OK;
  }
#endif
  if (skg_gemm8_try_launch(p, MODE, st)) {
    SKG_CHECK_LAUNCH("skg_gemm (v8)");
    return SKG_OK;
  }
#ifdef SKG_LAB      // withdrawn round-3 experiment (tools/lab/gemmk.hip, EXPERIMENTS.md): lab build only, SKG_GEMMK=1
  if (skg_gemmk_try_launch(p, MODE, st)) {
    SKG_CHECK_LAUNCH("skg_gemm (k-pair)");
    return SKG_OK;
  }
#endif
  if (skg_gemm2_try_launch(p, MODE, st)) {
    SKG_CHECK_LAUNCH("skg_gemm (v2)");
    return SKG_OK;
  }
  if (p.flags & SKG_EPI_GEGLU) return SKG_E_UNSUPPORTED;     // fused GEGLU exists in the LDS-DMA kernel only
  if (p.c_lo || p.res_lo) return SKG_E_UNSUPPORTED;          // so does the hi / lo epilogue
  if (p.ntaps || p.up2 || p.seg_rows || p.K2) return SKG_E_UNSUPPORTED;      // and the polyphase tap walk / the output row maps / the second operand
  const int tm = skg_cdiv(p.M, BM);
  if (use_wide(p.M, p.N)) {
    dim3 grid(p.N / 128, tm);
    hipLaunchKernelGGL((gemm_kernel<128, MODE>), grid, dim3(256), 0, st, p);
  } else {
    dim3 grid(skg_cdiv(p.N, 64), tm);
    hipLaunchKernelGGL((gemm_kernel<64, MODE>), grid, dim3(256), 0, st, p);
  }
  SKG_CHECK_LAUNCH("skg_gemm");
  return SKG_OK;
}

}  // namespace

// ---- split-K workspaces: one caller-owned slab per (device, stream) ---------------------------------------------------
// Launches on ONE stream are ordered and may share a slab; launches on different streams (several pipelines in one process, a
// capture stream beside the eager stream) may run concurrently and must not.  The registry is read under a mutex by every
// gemm / conv entry point and the slab pointer travels to the kernel as an argument, so two host threads on two streams
// never see each other's slab (round 2 kept ONE process-global pointer that the Python side re-pointed: ADVICE r2).
namespace {
struct WsEntry { int dev; hipStream_t st; float* ws; size_t bytes; };
std::mutex g_ws_mu;
std::vector<WsEntry> g_ws_tab;
size_t g_ws_last_bytes = 0;        // what the shape queries (skg_gemm_variant, skg_gemm_gn_fused) assume; guarded by g_ws_mu

size_t ws_query_bytes() {          // the slab size the shape queries plan with: the most recent registration still standing, 0 with none
  std::lock_guard<std::mutex> lk(g_ws_mu);
  return g_ws_last_bytes;
}

void ws_attach(GemmParams& p, hipStream_t st) {
  int dev = 0;
  (void)hipGetDevice(&dev);
  std::lock_guard<std::mutex> lk(g_ws_mu);
  for (const WsEntry& e : g_ws_tab)
    if (e.dev == dev && e.st == st) {
#ifdef SKG_LAB      // (the withdrawn self-finishing split-K launch keeps its tile tickets at the end of the slab)
      p.ws = e.ws; p.ws_bytes = e.bytes - SKG_WS_TICKET_BYTES;
      p.ws_cnt = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(e.ws) + p.ws_bytes);
#else
      p.ws = e.ws; p.ws_bytes = e.bytes;
#endif
      return;
    }
  p.ws = nullptr; p.ws_bytes = 0; p.ws_cnt = nullptr;
}
}  // namespace

extern "C" int skg_set_workspace(void* ws, size_t bytes, void* stream) {
  SKG_REQUIRE(ws == nullptr || (skg_aligned(ws, 16) && bytes >= (1u << 20)));
  int dev = 0;
  (void)hipGetDevice(&dev);
#ifdef SKG_LAB
  if (ws) {      // the tile tickets of the self-finishing split-K launches start at zero (every launch leaves them there)
    if (hipMemsetAsync(reinterpret_cast<char*>(ws) + bytes - SKG_WS_TICKET_BYTES, 0, SKG_WS_TICKET_BYTES, (hipStream_t)stream) != hipSuccess) {
      (void)hipGetLastError();
      return SKG_E_LAUNCH;
    }
  }
#endif
  std::lock_guard<std::mutex> lk(g_ws_mu);
  for (size_t i = 0; i < g_ws_tab.size(); ++i)
    if (g_ws_tab[i].dev == dev && g_ws_tab[i].st == (hipStream_t)stream) {
      if (ws) { g_ws_tab[i].ws = (float*)ws; g_ws_tab[i].bytes = bytes; g_ws_last_bytes = bytes; }
      else {
        g_ws_tab.erase(g_ws_tab.begin() + i);
        g_ws_last_bytes = g_ws_tab.empty() ? 0 : g_ws_tab.back().bytes;      // (a query must not plan with a slab nobody holds any more)
      }
      return SKG_OK;
    }
  if (ws) { g_ws_tab.push_back({dev, (hipStream_t)stream, (float*)ws, bytes}); g_ws_last_bytes = bytes; }
  return SKG_OK;
}

extern "C" int skg_gemm_variant(int M, int N, int K, int Cin, int mode) {
  {   // v8 takes plain (fp16 out, no fused GEGLU) launches of eligible shapes
    GemmParams q{};
    q.M = M; q.N = N; q.K = K; q.Cin = Cin; q.lda = q.ldb = K; q.ldc = N; q.OH = q.OW = q.IH = q.IW = 1;
#ifdef SKG_LAB
    if (skg_gemmws_eligible(q, mode)) return 7320;      // (plain epilogue only: launches with statistics / GEGLU / fp32 out take v2)
#endif
#ifdef SKG_LAB
    q.C = (void*)16;      // (alignment checks only)
    if (skg_gemm9_eligible(q, mode)) return 9320;
    q.C = nullptr;
#endif
    if (const int bn8 = skg_gemm8_tile_n(q, mode)) return 8000 + bn8;
#ifdef SKG_LAB
    q.C = (void*)16;      // (alignment checks only)
    if (skg_gemmk_eligible(q, mode)) return 9160;
#endif
  }
  const int v2 = skg_gemm2_tile_n(M, N, K, Cin, mode, ws_query_bytes());
  return v2 ? 2000 + v2 : 1000 + (use_wide(M, N) ? 128 : 64);
}

static int gemm_impl(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                     const void* bias, const void* residual, int ldr, float alpha, unsigned flags, float* gn_partial,
                     int HW, int groups, void* stream, void* c_lo = nullptr, const void* res_lo = nullptr);
static int conv_impl(const void* X, int ldx, const void* Wp, void* Y, int ldy, int rows, int IH, int IW, int Cin,
                     int Cout, int mode, const void* bias, const void* residual, int ldr, float alpha, unsigned flags,
                     float* gn_partial, int groups, void* stream, void* c_lo = nullptr, const void* res_lo = nullptr);

extern "C" int skg_gemm_f16_geglu_keep(const void* A, int lda, const void* B, int ldb, void* Y, int ldy, void* H,
                                       int ldh, int M, int N, int K, const void* bias, void* stream) {
  SKG_REQUIRE(A && B && Y && H && M > 0 && N > 0 && K > 0 && K % 64 == 0 && N % 16 == 0);
  SKG_REQUIRE(lda % 8 == 0 && ldb % 8 == 0 && ldy % 8 == 0 && ldh % 8 == 0 && lda >= K && ldb >= K && ldy >= N / 2 && ldh >= N);
  SKG_REQUIRE(skg_aligned(A, 16) && skg_aligned(B, 16) && skg_aligned(Y, 16) && skg_aligned(H, 16));
  SKG_REQUIRE(!bias || skg_aligned(bias, 8));
  GemmParams p{};
  p.A = (const half_t*)A; p.lda = lda; p.B = (const half_t*)B; p.ldb = ldb; p.C = Y; p.ldc = ldy;
  p.bias = (const half_t*)bias; p.M = M; p.N = N; p.K = K; p.alpha = 1.f; p.flags = SKG_EPI_GEGLU;
  p.aux = (half_t*)H; p.ldaux = ldh;
  ws_attach(p, (hipStream_t)stream);
  return launch<MODE_DIRECT>(p, (hipStream_t)stream);
}

// 1 when skg_gemm_f16_gn / skg_conv3x3_f16_gn of this shape (contiguous, 16-byte aligned output) gets its partial sums
// from the kernel's own epilogue, 0 when the stand-alone statistics pass follows (bench.py spells kernel names from it)
extern "C" int skg_gemm_gn_fused(int M, int N, int K, int Cin, int mode, int HW, int groups) {
  if (HW <= 0 || groups <= 0 || M <= 0 || N <= 0) return 0;
  static float dummy;
  GemmParams q{};
  q.M = M; q.N = N; q.K = K; q.Cin = Cin; q.lda = q.ldb = K; q.ldc = N; q.OH = q.OW = q.IH = q.IW = 1;
  q.gn_partial = &dummy; q.gn_hw = HW; q.gn_groups = groups;
  const size_t wsb = ws_query_bytes();
  q.ws = wsb ? (float*)&dummy : nullptr; q.ws_bytes = wsb;
#ifdef SKG_LAB
  if (skg_gemm9_eligible(q, mode)) return 0;
#endif
  return (skg_gemm8_eligible(q, mode) ? skg_gemm8_fuses_gn(q, mode) : skg_gemm2_fuses_gn(q, mode)) ? 1 : 0;
}

extern "C" int skg_gemm_f16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M,
                            int N, int K, const void* bias, const void* residual, int ldr,
                            float alpha, unsigned flags, void* stream) {
  return gemm_impl(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, alpha, flags, nullptr, 0, 0, stream);
}

extern "C" int skg_gemm_f16_rows(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                                 const void* bias, int seg_rows, int seg_stride, void* stream) {
  SKG_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0 && seg_rows > 0 && seg_stride >= seg_rows && M % seg_rows == 0);
  SKG_REQUIRE(K % 64 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0 && lda >= K && ldb >= K && ldc >= N);
  SKG_REQUIRE(skg_aligned(A, 16) && skg_aligned(B, 16) && skg_aligned(C, 16) && (!bias || skg_aligned(bias, 8)));
  GemmParams p{};
  p.A = (const half_t*)A; p.lda = lda; p.B = (const half_t*)B; p.ldb = ldb; p.C = C; p.ldc = ldc;
  p.bias = (const half_t*)bias; p.M = M; p.N = N; p.K = K; p.alpha = 1.f; p.flags = 0;
  p.seg_rows = seg_rows; p.seg_stride = seg_stride;
  ws_attach(p, (hipStream_t)stream);
  if (!skg_gemm2_try_launch(p, MODE_DIRECT, (hipStream_t)stream)) return SKG_E_UNSUPPORTED;
  SKG_CHECK_LAUNCH("skg_gemm_f16_rows");
  return SKG_OK;
}

static int gemm_impl(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                     const void* bias, const void* residual, int ldr, float alpha, unsigned flags, float* gn_partial,
                     int HW, int groups, void* stream, void* c_lo, const void* res_lo) {
  SKG_REQUIRE(A && B && C && M > 0 && N > 0 && K > 0);
  SKG_REQUIRE(K % 32 == 0 && N % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0 && ldc % 4 == 0);
  SKG_REQUIRE(skg_aligned(A, 16) && skg_aligned(B, 16) && skg_aligned(C, 8));
  SKG_REQUIRE(!bias || skg_aligned(bias, 8));
  SKG_REQUIRE(!residual || (skg_aligned(residual, 8) && ldr % 4 == 0));
  SKG_REQUIRE(lda >= K && ldb >= K && ldc >= ((flags & SKG_EPI_GEGLU) ? N / 2 : N));
  GemmParams p{};
  p.A = (const half_t*)A; p.lda = lda; p.B = (const half_t*)B; p.ldb = ldb; p.C = C; p.ldc = ldc;
  p.bias = (const half_t*)bias; p.res = (const half_t*)residual; p.ldr = ldr;
  p.M = M; p.N = N; p.K = K; p.alpha = alpha; p.flags = flags;
  p.gn_partial = gn_partial; p.gn_hw = HW; p.gn_groups = groups;
  p.c_lo = (half_t*)c_lo; p.res_lo = (const half_t*)res_lo;
  ws_attach(p, (hipStream_t)stream);
  return launch<MODE_DIRECT>(p, (hipStream_t)stream);
}

static bool gn_args_ok(const float* partial, int M, int N, int HW, int groups, int ldc, unsigned flags) {
  return partial && HW > 0 && HW % 128 == 0 && HW / 128 <= 128 && M % HW == 0 && groups > 0 && groups <= 64 &&
         N % groups == 0 && N % 8 == 0 && (N / groups) % 2 == 0 && (N / groups) >= 4 && N <= 4096 && ldc % 8 == 0 &&
         !(flags & (SKG_EPI_OUT_F32 | SKG_EPI_GEGLU));
}

extern "C" int skg_gemm_f16_gn(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                               const void* bias, const void* residual, int ldr, float alpha, unsigned flags,
                               float* gn_partial, int HW, int groups, void* stream) {
  SKG_REQUIRE(gn_args_ok(gn_partial, M, N, HW, groups, ldc, flags) && skg_aligned(C, 16));
  return gemm_impl(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, alpha, flags, gn_partial, HW, groups, stream);
}

extern "C" int skg_conv3x3_f16_gn(const void* X, int ldx, const void* Wp, void* Y, int ldy, int rows, int IH, int IW,
                                  int Cin, int Cout, int mode, const void* bias, const void* residual, int ldr,
                                  float alpha, unsigned flags, float* gn_partial, int groups, void* stream) {
  const int up = (mode == SKG_CONV_UP2 || mode == SKG_CONV_S2T), dn = (mode == SKG_CONV_S2 || mode == SKG_CONV_S2A);
  const int OH = up ? IH * 2 : dn ? IH / 2 : IH, OW = up ? IW * 2 : dn ? IW / 2 : IW;
  SKG_REQUIRE(rows > 0 && OH > 0 && OW > 0);
  SKG_REQUIRE(gn_args_ok(gn_partial, rows * OH * OW, Cout, OH * OW, groups, ldy, flags) && skg_aligned(Y, 16));
  return conv_impl(X, ldx, Wp, Y, ldy, rows, IH, IW, Cin, Cout, mode, bias, residual, ldr, alpha, flags, gn_partial,
                   groups, stream);
}

// ---- conv2 + conv_shortcut of a ResnetBlock as ONE implicit GEMM (include/skg.h) -------------------------------------------
extern "C" int skg_conv3x3_sc_f16(const void* X, int ldx, const void* X2, int ldx2, int K2, const void* Wcat, void* Y, void* Y_lo,
                                  int ldy, int rows, int IH, int IW, int Cin, int Cout, const void* bias, unsigned flags,
                                  float* gn_partial, int groups, void* stream) {
  SKG_REQUIRE(X && X2 && Wcat && Y && rows > 0 && IH > 0 && IW > 0 && K2 > 0 && K2 % 64 == 0 && Cin % 64 == 0 && Cout % 8 == 0);
  SKG_REQUIRE(ldx % 8 == 0 && ldx >= Cin && ldx2 % 8 == 0 && ldx2 >= K2 && ldy % 8 == 0 && ldy >= Cout);
  SKG_REQUIRE(skg_aligned(X, 16) && skg_aligned(X2, 16) && skg_aligned(Wcat, 16) && skg_aligned(Y, 16) && (!Y_lo || skg_aligned(Y_lo, 16)) &&
              (!bias || skg_aligned(bias, 8)) && !(flags & (SKG_EPI_OUT_F32 | SKG_EPI_GEGLU)));
  // the second operand goes through a 32-bit buffer descriptor: a larger one is DECLINED (nothing launched; the caller runs
  // conv2 and the shortcut GEMM as two launches), not an argument error
  if ((unsigned long long)rows * IH * IW * ldx2 * 2ull >= 0x7fffffffull) return SKG_E_UNSUPPORTED;
  GemmParams p{};
  p.A = (const half_t*)X; p.lda = ldx; p.B = (const half_t*)Wcat; p.ldb = 9 * Cin + K2; p.C = Y; p.ldc = ldy;
  p.bias = (const half_t*)bias; p.c_lo = (half_t*)Y_lo;
  p.N = Cout; p.K = 9 * Cin + K2; p.alpha = 1.f; p.flags = flags;
  p.IH = IH; p.IW = IW; p.Cin = Cin; p.OH = IH; p.OW = IW; p.M = rows * IH * IW; p.gn_hw = IH * IW;
  p.A2 = (const half_t*)X2; p.lda2 = ldx2; p.K2 = K2;
  if (gn_partial) {
    SKG_REQUIRE(gn_args_ok(gn_partial, p.M, Cout, IH * IW, groups, ldy, flags));
    p.gn_partial = gn_partial; p.gn_groups = groups;
  }
  hipStream_t st = (hipStream_t)stream;
  ws_attach(p, st);
  return launch<MODE_S1>(p, st);
}

// ---- nearest-2x upsample + 3x3 conv, polyphase (include/skg.h) -----------------------------------------------------------
// Output pixel (2 i + a, 2 j + b) only ever reads the 2 x 2 low-resolution neighbourhood rows {i - 1, i} (a = 0) or {i, i + 1}
// (a = 1) x the same for columns, with the three taps of the 3 x 3 filter that land on one low-res row / column pre-summed:
// four stride-1 convolutions with FOUR taps each over the low-res input (16 tap-products per low-res pixel) instead of nine
// taps at every high-res pixel (36).  Wpp: [4 phases (2 a + b)][Cout][4 taps (row-major over the phase's 2 x 2)][Cin].
static int conv_up2_impl(const void* X, int ldx, const void* Wpp, void* Y, void* Y_lo, int ldy, int rows, int IH, int IW, int Cin,
                         int Cout, int a_wrap, const void* bias, void* stream) {
  SKG_REQUIRE(X && Wpp && Y && rows > 0 && IH > 0 && IW > 0 && Cin % 64 == 0 && Cout % 8 == 0);
  SKG_REQUIRE(ldx % 8 == 0 && ldx >= (a_wrap ? a_wrap : Cin) && ldy % 8 == 0 && ldy >= Cout && skg_aligned(X, 16) && skg_aligned(Wpp, 16) &&
              skg_aligned(Y, 16) && skg_aligned(Y_lo, 16) && (!bias || skg_aligned(bias, 8)));
  hipStream_t st = (hipStream_t)stream;
  // a phase that does not fill the chip by itself (the 8 -> 16 / 12 -> 24 maps): the four phases as ONE grid
  const bool one_grid = (long)skg_cdiv(rows * IH * IW, 128) * skg_cdiv(Cout, 160) < 200;
  for (int ph = 0; ph < (one_grid ? 1 : 4); ++ph) {
    const int a = ph >> 1, b = ph & 1;
    GemmParams p{};
    p.A = (const half_t*)X; p.lda = ldx;
    p.B = (const half_t*)Wpp + (size_t)ph * Cout * 4 * Cin; p.ldb = 4 * Cin;
    p.C = Y; p.ldc = ldy; p.bias = (const half_t*)bias; p.c_lo = (half_t*)Y_lo;
    p.N = Cout; p.K = 4 * Cin; p.alpha = 1.f; p.flags = 0;
    p.IH = IH; p.IW = IW; p.OH = IH; p.OW = IW; p.Cin = Cin; p.M = rows * IH * IW;
    p.ntaps = 4; p.up2 = one_grid ? 5 : 1 + ph;
    p.a_wrap = a_wrap;
    // low-res rows (ky) / columns (kx) this phase reads, as stride-1 tap ids ky * 3 + kx with ky, kx in {0: -1, 1: 0, 2: +1}
    const int ky0 = a ? 1 : 0, kx0 = b ? 1 : 0;
    p.tapmap = (unsigned)(ky0 * 3 + kx0) | (unsigned)(ky0 * 3 + kx0 + 1) << 4 | (unsigned)((ky0 + 1) * 3 + kx0) << 8 |
               (unsigned)((ky0 + 1) * 3 + kx0 + 1) << 12;
    ws_attach(p, st);
    if (!skg_gemm2_try_launch(p, MODE_S1, st)) return SKG_E_UNSUPPORTED;
    SKG_CHECK_LAUNCH("skg_conv3x3_up2_f16");
  }
  return SKG_OK;
}

// ---- 3x3 stride-1 convolution by Winograd F(2x2, 3x3) (include/skg.h, wino.hip) ---------------------------------------------
extern "C" size_t skg_conv3x3_wino_v_bytes(int rows, int IH, int IW, int Cin) {
  if (rows <= 0 || IH <= 0 || IW <= 0 || Cin <= 0) return 0;
  return (size_t)rows * (IH / 2) * (IW / 2) * 16 * Cin * 2;
}

extern "C" int skg_conv3x3_wino_f16(const void* X, int ldx, const void* U, void* V, void* Y, void* Y_lo, int ldy, int rows, int IH, int IW,
                                    int Cin, int Cout, const void* bias, const void* residual, const void* residual_lo, int ldr,
                                    unsigned flags, void* stream) {
  SKG_REQUIRE(U && V && Y && rows > 0 && IH > 0 && IW > 0 && Cin > 0 && Cout > 0);
  SKG_REQUIRE((!X || (ldx % 8 == 0 && ldx >= Cin)) && ldy % 4 == 0 && ldy >= Cout && skg_aligned(X, 16) && skg_aligned(U, 16) && skg_aligned(V, 16) &&
              skg_aligned(Y, 8) && (!Y_lo || skg_aligned(Y_lo, 8)) && (!bias || skg_aligned(bias, 8)));
  SKG_REQUIRE((!residual && !residual_lo) || (ldr % 4 == 0 && ldr >= Cout && (!residual || skg_aligned(residual, 8)) &&
                                              (!residual_lo || skg_aligned(residual_lo, 8))));
  SKG_REQUIRE(!(flags & ~SKG_EPI_RELU));
  // shapes this path takes (everything else: SKG_E_UNSUPPORTED, nothing launched - run skg_conv3x3_f16): even maps, whole 64-deep K
  // tiles per transform component, four output columns per thread of the output transform
  if ((IH & 1) || (IW & 1) || Cin % 64 != 0 || Cout % 8 != 0) return SKG_E_UNSUPPORTED;
  hipStream_t st = (hipStream_t)stream;
  GemmParams p{};
  p.A = (const half_t*)V; p.lda = 16 * Cin; p.B = (const half_t*)U; p.ldb = 16 * Cin; p.C = Y; p.ldc = ldy;
  p.bias = (const half_t*)bias; p.res = (const half_t*)residual; p.res_lo = (const half_t*)residual_lo; p.ldr = ldr;
  p.c_lo = (half_t*)Y_lo;
  p.M = rows * (IH / 2) * (IW / 2); p.N = Cout; p.K = 16 * Cin; p.alpha = 1.f; p.flags = flags;
  p.wino = 16;
  ws_attach(p, st);
  if (!p.ws || (size_t)16 * p.M * p.N * 4 > p.ws_bytes) return SKG_E_UNSUPPORTED;      // the 16 fp32 slabs live in the stream's workspace
  {   // (the launcher only reads the shapes and pointers of the GEMM proper: the epilogue fields belong to the output transform)
    unsigned long long vb = (unsigned long long)p.M * p.lda * 2ull, ub = (unsigned long long)p.N * p.ldb * 2ull;
    if (vb >= 0x7fffffffull || ub >= 0x7fffffffull) return SKG_E_UNSUPPORTED;
  }
  if (X) {      // (X == NULL: V already holds the input transform - skg_groupnorm_wino_fwd wrote it)
    skg_wino_in_launch((const half_t*)X, ldx, (half_t*)V, rows, IH, IW, Cin, st);
    SKG_CHECK_LAUNCH("skg_conv3x3_wino_f16 (input transform)");
  }
  GemmParams g = p;      // the GEMM step: raw slabs only
  g.bias = nullptr; g.res = nullptr; g.res_lo = nullptr; g.c_lo = nullptr; g.flags = 0;
  if (!skg_gemm2_try_launch(g, MODE_DIRECT, st)) return SKG_E_UNSUPPORTED;
  SKG_CHECK_LAUNCH("skg_conv3x3_wino_f16 (GEMM)");
  skg_wino_out_launch(p, p.ws, IH, IW, st);
  SKG_CHECK_LAUNCH("skg_conv3x3_wino_f16 (output transform)");
  return SKG_OK;
}

extern "C" int skg_conv3x3_up2_f16(const void* X, int ldx, const void* Wpp, void* Y, int ldy, int rows, int IH, int IW,
                                   int Cin, int Cout, const void* bias, void* stream) {
  return conv_up2_impl(X, ldx, Wpp, Y, nullptr, ldy, rows, IH, IW, Cin, Cout, 0, bias, stream);
}

// Accuracy mode: the input is the pair buffer [x_hi | x_lo] (2 C channels per pixel, pitch ldx), the pre-summed polyphase weights
// are (hi, lo) pairs too - their fp16 rounding would otherwise cost the margin of the eps bound (EXPERIMENTS.md round 3) - and a
// tap's K axis is [x_hi | x_lo | x_hi] . [W_hi | W_hi | W_lo] (x_lo . W_lo is dropped: 2^-22): Wpp3 [4][Cout][4 taps][3 C];
// the output is the pair Y + Y_lo.  12 tap-products of C channels per low-res pixel and phase against 18 for the 9-tap pair form.
extern "C" int skg_conv3x3_up2_f16_hilo(const void* X2, int ldx, const void* Wpp3, void* Y, void* Y_lo, int ldy, int rows, int IH,
                                        int IW, int C, int Cout, const void* bias, void* stream) {
  SKG_REQUIRE(Y_lo && C % 64 == 0);
  return conv_up2_impl(X2, ldx, Wpp3, Y, Y_lo, ldy, rows, IH, IW, 3 * C, Cout, 2 * C, bias, stream);
}

// Accuracy mode, round 5: the polyphase launch of skg_conv3x3_up2_f16 on the hi part of the stream with a PAIR output.  Of the three
// K-thirds of the form above, x_lo W_hi and x_hi W_lo each move eps by ~1 % of the mode's distance from fp32 (tools/eps_decompose_up.py:
// rel 5.00e-4 -> 5.18e-4 with both dropped) and cost two thirds of the upsamplers' time, the largest single item of the mode's price.
extern "C" int skg_conv3x3_up2_f16_pairout(const void* X, int ldx, const void* Wpp, void* Y, void* Y_lo, int ldy, int rows, int IH,
                                           int IW, int Cin, int Cout, const void* bias, void* stream) {
  SKG_REQUIRE(Y_lo);
  return conv_up2_impl(X, ldx, Wpp, Y, Y_lo, ldy, rows, IH, IW, Cin, Cout, 0, bias, stream);
}

// dX of the polyphase upsample + conv above = ONE 4 x 4 stride-2 convolution (padding 1) over dY at the upsampled size with
// the transposed pre-summed weights: dX[i, j] = sum_{ky, kx < 4} W16[ky, kx]^T dY[2 i - 1 + ky, 2 j - 1 + kx] - 16 tap-products
// per low-res pixel where the 9-tap dgrad at the upsampled size + 2 x 2 sum-pool spends 36.
extern "C" int skg_conv4x4s2_f16(const void* X, int ldx, const void* W16, void* Y, int ldy, int rows, int IH, int IW,
                                 int Cin, int Cout, const void* bias, void* stream) {
  SKG_REQUIRE(X && W16 && Y && rows > 0 && IH > 0 && IW > 0 && IH % 2 == 0 && IW % 2 == 0 && Cin % 64 == 0 && Cout % 8 == 0);
  SKG_REQUIRE(ldx % 8 == 0 && ldx >= Cin && ldy % 8 == 0 && ldy >= Cout && skg_aligned(X, 16) && skg_aligned(W16, 16) &&
              skg_aligned(Y, 16) && (!bias || skg_aligned(bias, 8)));
  hipStream_t st = (hipStream_t)stream;
  GemmParams p{};
  p.A = (const half_t*)X; p.lda = ldx;
  p.B = (const half_t*)W16; p.ldb = 16 * Cin;
  p.C = Y; p.ldc = ldy; p.bias = (const half_t*)bias;
  p.N = Cout; p.K = 16 * Cin; p.alpha = 1.f; p.flags = 0;
  p.IH = IH; p.IW = IW; p.OH = IH / 2; p.OW = IW / 2; p.Cin = Cin; p.M = rows * p.OH * p.OW;
  p.ntaps = 16;
  ws_attach(p, st);
  if (!skg_gemm2_try_launch(p, MODE_S2, st)) return SKG_E_UNSUPPORTED;
  SKG_CHECK_LAUNCH("skg_conv4x4s2_f16");
  return SKG_OK;
}

// ---- accuracy mode: outputs / residuals as (hi, lo) pairs of fp16 tensors (include/skg.h) ------------------------------
extern "C" int skg_gemm_f16_hilo(const void* A, int lda, const void* B, int ldb, void* C, void* C_lo, int ldc, int M, int N,
                                 int K, const void* bias, const void* residual, const void* residual_lo, int ldr,
                                 float alpha, unsigned flags, void* stream) {
  SKG_REQUIRE((C_lo || residual_lo) && !(flags & (SKG_EPI_OUT_F32 | SKG_EPI_GEGLU)) && K % 64 == 0 && ldc % 8 == 0);
  SKG_REQUIRE(skg_aligned(C, 16) && (!C_lo || skg_aligned(C_lo, 16)) && (!residual_lo || skg_aligned(residual_lo, 16)) &&
              (!residual || skg_aligned(residual, 16)) && ((!residual && !residual_lo) || ldr % 8 == 0));
  return gemm_impl(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, alpha, flags, nullptr, 0, 0, stream, C_lo, residual_lo);
}

extern "C" int skg_conv3x3_f16_hilo(const void* X, int ldx, const void* Wp, void* Y, void* Y_lo, int ldy, int rows, int IH,
                                    int IW, int Cin, int Cout, int mode, const void* bias, const void* residual,
                                    const void* residual_lo, int ldr, float alpha, unsigned flags, void* stream) {
  SKG_REQUIRE((Y_lo || residual_lo) && !(flags & (SKG_EPI_OUT_F32 | SKG_EPI_GEGLU)) && Cin % 64 == 0 && ldy % 8 == 0);
  SKG_REQUIRE(skg_aligned(Y, 16) && (!Y_lo || skg_aligned(Y_lo, 16)) && (!residual_lo || skg_aligned(residual_lo, 16)) &&
              (!residual || skg_aligned(residual, 16)) && ((!residual && !residual_lo) || ldr % 8 == 0));
  return conv_impl(X, ldx, Wp, Y, ldy, rows, IH, IW, Cin, Cout, mode, bias, residual, ldr, alpha, flags, nullptr, 0, stream,
                   Y_lo, residual_lo);
}

// ... and the GroupNorm partial sums of the OUTPUT's hi part with it (what skg_groupnorm_from_partial_hilo folds): from the
// epilogue of the kernel that runs where its tile can (256 x 320, 128 x 160), else from the stand-alone pass
extern "C" int skg_gemm_f16_hilo_gn(const void* A, int lda, const void* B, int ldb, void* C, void* C_lo, int ldc, int M, int N,
                                    int K, const void* bias, const void* residual, const void* residual_lo, int ldr,
                                    float alpha, unsigned flags, float* gn_partial, int HW, int groups, void* stream) {
  SKG_REQUIRE((C_lo || residual_lo) && !(flags & (SKG_EPI_OUT_F32 | SKG_EPI_GEGLU)) && K % 64 == 0 && ldc % 8 == 0);
  SKG_REQUIRE(skg_aligned(C, 16) && (!C_lo || skg_aligned(C_lo, 16)) && (!residual_lo || skg_aligned(residual_lo, 16)) &&
              (!residual || skg_aligned(residual, 16)) && ((!residual && !residual_lo) || ldr % 8 == 0));
  SKG_REQUIRE(gn_args_ok(gn_partial, M, N, HW, groups, ldc, flags));
  return gemm_impl(A, lda, B, ldb, C, ldc, M, N, K, bias, residual, ldr, alpha, flags, gn_partial, HW, groups, stream, C_lo,
                   residual_lo);
}

extern "C" int skg_conv3x3_f16_hilo_gn(const void* X, int ldx, const void* Wp, void* Y, void* Y_lo, int ldy, int rows, int IH,
                                       int IW, int Cin, int Cout, int mode, const void* bias, const void* residual,
                                       const void* residual_lo, int ldr, float alpha, unsigned flags, float* gn_partial,
                                       int groups, void* stream) {
  SKG_REQUIRE((Y_lo || residual_lo) && !(flags & (SKG_EPI_OUT_F32 | SKG_EPI_GEGLU)) && Cin % 64 == 0 && ldy % 8 == 0);
  SKG_REQUIRE(skg_aligned(Y, 16) && (!Y_lo || skg_aligned(Y_lo, 16)) && (!residual_lo || skg_aligned(residual_lo, 16)) &&
              (!residual || skg_aligned(residual, 16)) && ((!residual && !residual_lo) || ldr % 8 == 0));
  const int up = (mode == SKG_CONV_UP2 || mode == SKG_CONV_S2T), dn = (mode == SKG_CONV_S2 || mode == SKG_CONV_S2A);
  const int OH = up ? IH * 2 : dn ? IH / 2 : IH, OW = up ? IW * 2 : dn ? IW / 2 : IW;
  SKG_REQUIRE(rows > 0 && OH > 0 && OW > 0 && gn_args_ok(gn_partial, rows * OH * OW, Cout, OH * OW, groups, ldy, flags));
  return conv_impl(X, ldx, Wp, Y, ldy, rows, IH, IW, Cin, Cout, mode, bias, residual, ldr, alpha, flags, gn_partial, groups,
                   stream, Y_lo, residual_lo);
}

extern "C" int skg_conv3x3_f16(const void* X, int ldx, const void* Wp, void* Y, int ldy, int rows,
                               int IH, int IW, int Cin, int Cout, int mode, const void* bias,
                               const void* residual, int ldr, float alpha, unsigned flags,
                               void* stream) {
  return conv_impl(X, ldx, Wp, Y, ldy, rows, IH, IW, Cin, Cout, mode, bias, residual, ldr, alpha, flags, nullptr, 0, stream);
}

static int conv_impl(const void* X, int ldx, const void* Wp, void* Y, int ldy, int rows, int IH, int IW, int Cin,
                     int Cout, int mode, const void* bias, const void* residual, int ldr, float alpha, unsigned flags,
                     float* gn_partial, int groups, void* stream, void* c_lo, const void* res_lo) {
  SKG_REQUIRE(X && Wp && Y && rows > 0 && IH > 0 && IW > 0);
  SKG_REQUIRE(Cin % 32 == 0 && Cout % 8 == 0 && ldx % 8 == 0 && ldx >= Cin && ldy % 4 == 0 && ldy >= Cout);
  SKG_REQUIRE(skg_aligned(X, 16) && skg_aligned(Wp, 16) && skg_aligned(Y, 8));
  SKG_REQUIRE(!bias || skg_aligned(bias, 8));
  SKG_REQUIRE(!residual || (skg_aligned(residual, 8) && ldr % 4 == 0));
  GemmParams p{};
  p.A = (const half_t*)X; p.lda = ldx; p.B = (const half_t*)Wp; p.ldb = 9 * Cin; p.C = Y; p.ldc = ldy;
  p.bias = (const half_t*)bias; p.res = (const half_t*)residual; p.ldr = ldr;
  p.N = Cout; p.K = 9 * Cin; p.alpha = alpha; p.flags = flags;
  p.IH = IH; p.IW = IW; p.Cin = Cin;
  p.gn_partial = gn_partial; p.gn_groups = groups;
  p.c_lo = (half_t*)c_lo; p.res_lo = (const half_t*)res_lo;
  hipStream_t st = (hipStream_t)stream;
  ws_attach(p, st);
  switch (mode) {
    case SKG_CONV_S1:
      p.OH = IH; p.OW = IW; p.M = rows * p.OH * p.OW; p.gn_hw = p.OH * p.OW;
      return launch<MODE_S1>(p, st);
    case SKG_CONV_S2:
      SKG_REQUIRE(IH % 2 == 0 && IW % 2 == 0);
      p.OH = IH / 2; p.OW = IW / 2; p.M = rows * p.OH * p.OW; p.gn_hw = p.OH * p.OW;
      return launch<MODE_S2>(p, st);
    case SKG_CONV_S2A:
      SKG_REQUIRE(IH % 2 == 0 && IW % 2 == 0);
      p.OH = IH / 2; p.OW = IW / 2; p.M = rows * p.OH * p.OW; p.gn_hw = p.OH * p.OW;
      return launch<MODE_S2A>(p, st);
    case SKG_CONV_UP2:
      p.OH = IH * 2; p.OW = IW * 2; p.M = rows * p.OH * p.OW; p.gn_hw = p.OH * p.OW;
      return launch<MODE_UP2>(p, st);
    case SKG_CONV_S2T:
      p.OH = IH * 2; p.OW = IW * 2; p.M = rows * p.OH * p.OW; p.gn_hw = p.OH * p.OW;
      return launch<MODE_S2T>(p, st);
    default:
      return SKG_E_UNSUPPORTED;
  }
}

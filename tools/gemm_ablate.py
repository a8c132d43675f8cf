"""Ablation of gemm2_kernel on conv shapes: full kernel vs DMA-only vs (MFMA + LDS reads)-only loops."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import ops
from sketch2img_amd._lib import lib, check
DEV = "cuda:0"


def run(cin, cout, hw, rows, flags, iters=20):
    x = torch.randn(rows * hw * hw, cin, device=DEV).half()
    w = (torch.randn(cout, 9 * cin, device=DEV) * (9 * cin) ** -0.5).half()
    out = torch.empty(rows * hw * hw, cout, device=DEV, dtype=torch.float16)
    st = ops._stream()
    f = lambda: check(lib.skg_conv3x3_f16(x.data_ptr(), cin, w.data_ptr(), out.data_ptr(), cout, rows, hw, hw, cin,
                                          cout, 0, None, None, 0, 1.0, flags, st), "conv")
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for (cin, cout, hw) in ((320, 320, 64), (960, 320, 64), (1280, 640, 32), (1280, 1280, 16)):
    t = [run(cin, cout, hw, 16, fl) for fl in (0, 0x100, 0x200)]
    fl = 2.0 * 16 * hw * hw * cout * 9 * cin
    print(f"conv {cin}->{cout}@{hw}: full {t[0]:.1f} us ({fl / t[0] / 1e6:.0f} TF/s) | DMA only {t[1]:.1f} us | "
          f"MFMA+LDS only {t[2]:.1f} us ({fl / t[2] / 1e6:.0f} TF/s)")


def run_gemm(M, N, K, flags, iters=20):
    a = torch.randn(M, K, device=DEV).half()
    w = (torch.randn(N, K, device=DEV) * K ** -0.5).half()
    b = torch.randn(N, device=DEV).half()
    out = torch.empty(M, N, device=DEV, dtype=torch.float16)
    st = ops._stream()
    f = lambda: check(lib.skg_gemm_f16(a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, b.data_ptr(), None, 0,
                                       1.0, flags, st), "gemm")
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for (M, N, K) in ((65536, 320, 320), (65536, 960, 320), (65536, 2560, 320), (16384, 640, 640), (16384, 5120, 640), (65536, 320, 1280)):
    t = [run_gemm(M, N, K, fl) for fl in (0, 0x400, 0x500, 0x600, 0x700)]
    fl = 2.0 * M * N * K
    print(f"gemm {M}x{N}x{K}: full {t[0]:.1f} us ({fl / t[0] / 1e6:.0f} TF/s) | no-epilogue {t[1]:.1f} | DMA only {t[2]:.1f} | "
          f"MFMA+LDS only {t[3]:.1f} | launch+prologue only {t[4]:.1f}")

print("--- epilogue only (no DMA, no MFMA) vs full")
for (M, N, K) in ((65536, 320, 320), (65536, 2560, 320), (16384, 5120, 640)):
    t = [run_gemm(M, N, K, fl) for fl in (0, 0x300, 0x700)]
    print(f"gemm {M}x{N}x{K}: full {t[0]:.1f} us | epilogue only {t[1]:.1f} us ({2.0 * M * N / t[1] / 1e6:.2f} TB/s written) | empty {t[2]:.1f} us")

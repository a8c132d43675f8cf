"""FF1 (GEGLU projection): GEMM + separate GEGLU kernel vs GEMM with the fused GEGLU epilogue."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import ops
DEV = "cuda:0"


def t(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for M, C in ((65536, 320), (16384, 640), (4096, 1280)):
    x = torch.randn(M, C, device=DEV).half()
    w = (torch.randn(8 * C, C, device=DEV) * C ** -0.5).half()
    b = torch.randn(8 * C, device=DEV).half()
    f = torch.empty(M, 8 * C, device=DEV, dtype=torch.float16)
    y = torch.empty(M, 4 * C, device=DEV, dtype=torch.float16)
    t_g = t(lambda: ops.gemm(x, w, f, bias=b))
    t_e = t(lambda: ops.geglu(f, y, interleaved=True))
    t_f = t(lambda: ops.gemm(x, w, y, bias=b, geglu=True))
    print(f"FF1 M={M} C={C}: gemm {t_g:.1f} us + geglu {t_e:.1f} us = {t_g + t_e:.1f} us | fused {t_f:.1f} us")

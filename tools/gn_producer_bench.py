"""Producer launches with / without the GroupNorm partial sums in their epilogue, and the GroupNorm that follows
(stand-alone statistics pass + apply vs. fold-and-apply from the producer's partial sums).  Cold-cache timing."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import ops
DEV = "cuda:0"
flush = torch.empty(1 << 28, device=DEV, dtype=torch.float32)


def timed(fn, iters=12):
    tot = 0.0
    for i in range(iters + 2):
        flush.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        if i >= 2:
            tot += e0.elapsed_time(e1) * 1e3
    return tot / iters


G = 32
for kind, rows, H, Ci, Co, res in (("conv", 16, 64, 320, 320, False), ("conv", 16, 64, 320, 320, True), ("conv", 16, 32, 640, 640, False),
                                   ("conv", 16, 32, 640, 640, True), ("gemm", 16, 64, 320, 320, True), ("gemm", 16, 32, 640, 640, True),
                                   ("conv", 8, 64, 320, 320, False), ("gemm", 8, 64, 320, 320, True)):
    M, HW = rows * H * H, H * H
    x = torch.randn(M, Ci, device=DEV).half()
    r = torch.randn(M, Co, device=DEV).half() if res else None
    b = torch.randn(Co, device=DEV).half()
    ga, be = torch.ones(Co, device=DEV).half(), torch.zeros(Co, device=DEV).half()
    y = torch.empty(M, Co, device=DEV, dtype=torch.float16)
    n = torch.empty(M, Co, device=DEV, dtype=torch.float16)
    if kind == "gemm":
        w = (torch.randn(Co, Ci, device=DEV) * Ci ** -0.5).half()
        plain = lambda: ops.gemm(x, w, y, bias=b, residual=r)
        fused = lambda: ops.gemm(x, w, y, bias=b, residual=r, gn_stats=(HW, G))
    else:
        w = (torch.randn(Co, 9 * Ci, device=DEV) * (9 * Ci) ** -0.5).half()
        plain = lambda: ops.conv3x3(x, w, rows, H, H, out=y, bias=b, residual=r)
        fused = lambda: ops.conv3x3(x, w, rows, H, H, out=y, bias=b, residual=r, gn_groups=G)
    _, part = fused()
    t_p, t_f = timed(plain), timed(fused)
    t_g0 = timed(lambda: ops.groupnorm(y, rows, HW, G, 1e-5, ga, be, True, out=n))
    t_g1 = timed(lambda: ops.groupnorm(y, rows, HW, G, 1e-5, ga, be, True, out=n, partial=part))
    print(f"{kind} rows{rows} {Ci}->{Co}@{H} res={int(res)}: producer {t_p:.1f} -> {t_f:.1f} us | GroupNorm {t_g0:.1f} -> {t_g1:.1f} us | "
          f"sum {t_p + t_g0:.1f} -> {t_f + t_g1:.1f}", flush=True)

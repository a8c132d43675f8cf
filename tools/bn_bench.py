"""LGP BatchNorm apply / ReLU+BN backward at the sampling shapes (8 samples x 2 segments x 4096 rows), cold cache."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import ops
DEV = "cuda:0"
flush = torch.empty(1 << 28, device=DEV, dtype=torch.float32)


def timed(fn, iters=10):
    tot = 0.0
    for i in range(iters + 2):
        flush.fill_(float(i))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record()
        torch.cuda.synchronize()
        if i >= 2:
            tot += e0.elapsed_time(e1) * 1e3
    return tot / iters


S, hw = 8, 4096
for C in (512, 256, 128, 64):
    x = torch.relu(torch.randn(2 * S * hw, C, device=DEV)).half()
    dy = torch.randn(2 * S * hw, C, device=DEV).half()
    ga, be = torch.ones(C, device=DEV).half(), torch.zeros(C, device=DEV).half()
    rm, rv = torch.zeros(C, device=DEV), torch.ones(C, device=DEV)
    st = ops.bn_stats(x, S, 2, hw, 1e-5, rm, rv)
    y = torch.empty_like(x)
    t0 = timed(lambda: ops.bn_apply(x, S, 2, hw, st, ga, be, out=y))
    t1 = timed(lambda: ops.bn_relu_bwd(x, dy, S, 2, hw, st, ga, True))
    mb = x.numel() * 2 / 1e6
    print(f"C={C}: bn_apply {t0:.1f} us ({2 * mb / t0 / 1e6 * 1e6 / 1e6:.2f} TB/s), relu+bn backward (3 launches) {t1:.1f} us", flush=True)

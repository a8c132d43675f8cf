"""FF1 (fused GEGLU) -> FF2 (+ residual) over the whole activation vs. in row chunks whose intermediate stays on die.

Every timed pair starts from a flushed Infinity Cache (a 1 GB fill between repeats, outside the timed span) - the
in-pipeline condition, not the warm-loop one."""
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


for M, C in ((65536, 320), (32768, 320), (16384, 640), (8192, 640), (4096, 1280)):
    x = torch.randn(M, C, device=DEV).half()
    r = torch.randn(M, C, device=DEV).half()
    w1 = (torch.randn(8 * C, C, device=DEV) * C ** -0.5).half()
    b1 = torch.randn(8 * C, device=DEV).half()
    w2 = (torch.randn(C, 4 * C, device=DEV) * (4 * C) ** -0.5).half()
    b2 = torch.randn(C, device=DEV).half()
    f = torch.empty(M, 4 * C, device=DEV, dtype=torch.float16)
    y = torch.empty(M, C, device=DEV, dtype=torch.float16)

    def run(chunks):
        step = M // chunks
        for c in range(chunks):
            s = slice(c * step, (c + 1) * step)
            ops.gemm(x[s], w1, f[s], bias=b1, geglu=True)
            ops.gemm(f[s], w2, y[s], bias=b2, residual=r[s])

    def run_small(chunks):          # the intermediate of every chunk lands in the SAME small buffer
        step = M // chunks
        for c in range(chunks):
            s = slice(c * step, (c + 1) * step)
            ops.gemm(x[s], w1, f[:step], bias=b1, geglu=True)
            ops.gemm(f[:step], w2, y[s], bias=b2, residual=r[s])

    ref = None
    line = [f"FF M={M} C={C}:"]
    for ch in (1, 2, 4, 8, 16):
        if M // ch < 2048:
            continue
        t = timed(lambda: run(ch))
        t2 = timed(lambda: run_small(ch)) if ch > 1 else t
        line.append(f"x{ch} {t:.0f}/{t2:.0f}")
    t_ff1 = timed(lambda: ops.gemm(x, w1, f, bias=b1, geglu=True))
    t_ff2 = timed(lambda: ops.gemm(f, w2, y, bias=b2, residual=r))
    line.append(f"| alone (cold): FF1 {t_ff1:.0f} FF2 {t_ff2:.0f} us")
    print(" ".join(line), flush=True)

"""HBM bandwidth probes: pure write (fill), pure copy, and skg_axpby copy, on buffers larger than the 256 MiB MALL."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import ops
DEV = "cuda:0"


def t(fn, iters=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


for mb in (42, 335, 1340):
    n = mb * 1024 * 1024 // 2
    a = torch.empty(n, device=DEV, dtype=torch.float16)
    b = torch.empty(n, device=DEV, dtype=torch.float16)
    tf = t(lambda: a.fill_(1.0))
    tc = t(lambda: b.copy_(a))
    a2, b2 = a.view(-1, 256), b.view(-1, 256)
    tk = t(lambda: ops.axpby(a2, out=b2))
    print(f"{mb:5d} MB: fill {mb / 1024 / tf / 1e3 * 1.0737:.2f} TB/s | copy (r+w) {2 * mb / 1024 / tc / 1e3 * 1.0737:.2f} TB/s | "
          f"skg_axpby copy {2 * mb / 1024 / tk / 1e3 * 1.0737:.2f} TB/s")

#!/usr/bin/env python3
"""gemmws.hip vs the tile kernel on the N = K = 320 shapes: warm loop (one buffer set) and cold (buffer sets rotating through
more than the 256 MB Infinity Cache).  SKG_GEMMWS is read per launch.
    make -C sketch2img_amd/csrc lab && SKG_LIB=sketch2img_amd/libskg_lab.so python tools/lab/gemmws_bench.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sketch2img_amd import ops

DEV = "cuda:0"
g = torch.Generator().manual_seed(1)
w = (torch.randn(320, 320, generator=g) * 320 ** -0.5).half().to(DEV)
b = torch.randn(320, generator=g).half().to(DEV)


def bench(M, res, nsets, iters=40):
    sets = []
    for _ in range(nsets):
        a = torch.randn(M, 320, device=DEV).half()
        r = torch.randn(M, 320, device=DEV).half() if res else None
        o = torch.empty(M, 320, device=DEV, dtype=torch.float16)
        sets.append((a, r, o))
    out = {}
    for ws in ("0", "1"):
        os.environ["SKG_GEMMWS"] = ws
        best = 1e9
        for _ in range(4):
            for i in range(4):
                a, r, o = sets[i % nsets]
                ops.gemm(a, w, out=o, bias=b, residual=r)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                a, r, o = sets[i % nsets]
                ops.gemm(a, w, out=o, bias=b, residual=r)
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) * 1e3 / iters)
        out[ws] = best
    return out


for M in (65536, 32768):
    for res in (False, True):
        for nsets, tag in ((1, "warm"), (10, "cold")):
            t = bench(M, res, nsets)
            mb = 2.0 * M * 320 * (3 if res else 2) / 1e6
            print(f"M{M} res={res} {tag}: tile kernel {t['0']:.1f} us, gemmws {t['1']:.1f} us   ({mb:.0f} MB: {mb / t['0']:.2f} / {mb / t['1']:.2f} TB/s)", flush=True)

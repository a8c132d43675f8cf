#!/usr/bin/env python3
"""Check of the withdrawn weight-stationary kernel (tools/lab/gemmws.hip; lab build: `make -C sketch2img_amd/csrc lab`):
    SKG_LIB=sketch2img_amd/libskg_lab.so python tools/lab/gemmws_check.py
vs an fp32 torch reference, bit-identity with the tile kernel, and two streams at once (the case that broke the first version:
counted vmcnt waits across stores)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SKG_GEMMWS"] = "1"
from sketch2img_amd import ops as _ops  # noqa: E402


def dev():
    return "cuda:0"


def check(ops):
    """gemmws.hip (N = K = 320, M >= 32768: the weights live in registers, activation rows stream through LDS): vs an fp32
    torch reference, and BIT-identical to the tile kernel - the same rows as two M = 16384 launches take gemm2.hip - with /
    without bias and residual, alpha != 1, strided views, a ragged last tile."""
    from sketch2img_amd._lib import lib
    g = torch.Generator().manual_seed(77)
    assert lib.skg_gemm_variant(65536, 320, 320, 0, 0) == 7320 and lib.skg_gemm_variant(16384, 320, 320, 0, 0) != 7320
    for M, bias, res, alpha in [(32768, True, True, 1.0), (65536, False, False, 1.0), (73728, True, False, 0.5), (32768 + 16384 - 24, True, True, 0.75)]:
        abuf = torch.randn(M, 328, generator=g).half().to(dev())
        a = abuf[:, 8:]                                                    # lda = 328
        w = (torch.randn(320, 320, generator=g) * 320 ** -0.5).half().to(dev())
        b = torch.randn(320, generator=g).half().to(dev()) if bias else None
        rbuf = torch.randn(M, 640, generator=g).half().to(dev()) if res else None
        r = rbuf[:, 320:] if res else None
        out = torch.zeros(M, 336, device=dev(), dtype=torch.float16)
        ops.gemm(a, w, out=out[:, 8:328], bias=b, residual=r, alpha=alpha)
        ref = alpha * (a.float() @ w.float().t() + (b.float() if bias else 0.0)) + (r.float() if res else 0.0)
        e = float((out[:, 8:328].float() - ref).norm() / ref.norm())
        stray = float(out[:, :8].abs().max() + out[:, 328:].abs().max())
        # the tile kernel on the same rows (launches below the streaming kernel's M threshold)
        tile = torch.empty(M, 320, device=dev(), dtype=torch.float16)
        for m0 in range(0, M, 16384):
            m1 = min(M, m0 + 16384)
            ops.gemm(a[m0:m1], w, out=tile[m0:m1], bias=b, residual=None if r is None else r[m0:m1], alpha=alpha)
        same = torch.equal(tile, out[:, 8:328])
        print(f"gemmws M{M} bias={bias} res={res} alpha={alpha}: rel {e:.2e}, stray {stray}, bit-identical to the tile kernel: {same}")
        assert e < 4e-4 and stray == 0 and same
    # two streams at once (a guided step's forked branch): the persistent kernel's counted waits must not depend on what
    # else the chip is doing - each stream's results equal its solo run, every repetition
    M = 65536
    prob = []
    for sidx in range(2):
        a = torch.randn(M, 320, generator=g).half().to(dev())
        w = (torch.randn(320, 320, generator=g) * 320 ** -0.5).half().to(dev())
        r = torch.randn(M, 320, generator=g).half().to(dev())
        prob.append((a, w, r, ops.gemm(a, w, residual=r), ops.gemm(a, w)))
    torch.cuda.synchronize()
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    for rep in range(6):
        got = []
        for sidx, st in enumerate(streams):
            with torch.cuda.stream(st):
                a, w, r, _, _ = prob[sidx]
                got.append([ops.gemm(a, w, residual=r) for _ in range(3)] + [ops.gemm(a, w) for _ in range(3)])
        torch.cuda.synchronize()
        for sidx in range(2):
            assert all(torch.equal(o, prob[sidx][3]) for o in got[sidx][:3]) and all(torch.equal(o, prob[sidx][4]) for o in got[sidx][3:])


if __name__ == "__main__":
    check(_ops)
    print("ok")

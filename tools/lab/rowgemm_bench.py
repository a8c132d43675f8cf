#!/usr/bin/env python3
"""skg_rowgemm_f16 (weights resident in LDS, rows through registers) against the gemm2.hip launches it replaces at the shapes of a
config-2 batch (K = 320, M = 65 536 / 32 768): attn1.to_out + residual, proj_in, the fused q / k / v projection with and without
norm1 in the same launch.  Every repetition works on another of --pool operand sets (the producer of X is another kernel in the
real batch: X comes from HBM / the Infinity Cache, not from L2).  WITHDRAWN experiment, lab build:
    make -C sketch2img_amd/csrc lab && SKG_LIB=$PWD/sketch2img_amd/libskg_lab.so python tools/lab/rowgemm_bench.py [--pool 6] [--reps 20]
    (SKG_RG_PROBE=1|2|3: no matrix work / no stores / no row loads on the + residual launches - wrong results)"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sketch2img_amd import ops  # noqa: E402
from sketch2img_amd._lib import LIB_PATH  # noqa: E402

import ctypes  # noqa: E402

_lab = ctypes.CDLL(LIB_PATH)      # the lab build: SKG_LIB=.../libskg_lab.so
P, I, F = ctypes.c_void_p, ctypes.c_int, ctypes.c_float
_lab.skg_rowgemm_f16.argtypes = [P, I, P, P, I, I, I, I, P, P, I, P]
_lab.skg_ln_rowgemm_f16.argtypes = [P, I, P, P, I, I, I, I, P, P, I, P, P, F, P, P]


def pack_rowgemm(w, dev):
    """[N, 320] -> fp16 [N / 16, 10, 512]: piece (u, ks): [lane = 16 g + l][i] = W[16 u + l][32 ks + 8 g + i]"""
    N, K = w.shape
    return w.detach().to("cpu", torch.float16).reshape(N // 16, 16, K // 32, 4, 8).permute(0, 2, 3, 1, 4).reshape(N // 16, K // 32, 512).contiguous().to(dev)


def rowgemm(X, Wpack, out, bias=None, residual=None, ln=None):
    p = lambda t: None if t is None else t.data_ptr()
    M, K = X.shape
    N = Wpack.shape[0] * 16
    st = torch.cuda.current_stream().cuda_stream
    ldr = residual.stride(0) if residual is not None else 0
    if ln is None:
        rc = _lab.skg_rowgemm_f16(p(X), X.stride(0), p(Wpack), p(out), out.stride(0), M, N, K, p(bias), p(residual), ldr, st)
    else:
        rc = _lab.skg_ln_rowgemm_f16(p(X), X.stride(0), p(Wpack), p(out), out.stride(0), M, N, K, p(bias), p(residual), ldr, p(ln[0]), p(ln[1]),
                                     ln[2], None, st)
    assert rc == 0, rc
    return out


ap = argparse.ArgumentParser()
ap.add_argument("--pool", type=int, default=6)
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
d = "cuda:0"
g = torch.Generator().manual_seed(3)


def timeit(fns):
    for f in fns:
        f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(args.reps):
            fns[i % len(fns)]()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / args.reps * 1e3)
    return best


for M, N, res, ln in [(65536, 320, True, False), (65536, 320, False, False), (65536, 960, False, False), (65536, 960, False, True),
                      (32768, 320, True, False), (32768, 960, False, True)]:
    K = 320
    w = (torch.randn(N, K, generator=g) * K ** -0.5).half()
    wd, wp = w.to(d), pack_rowgemm(w, d)
    b = torch.randn(N, generator=g).half().to(d)
    gam, bet = torch.ones(K).half().to(d), torch.zeros(K).half().to(d)
    xs = [torch.randn(M, K, generator=g).half().to(d) for _ in range(args.pool)]
    rs = [torch.randn(M, N, generator=g).half().to(d) if res else None for _ in range(args.pool)]
    outs = [torch.empty(M, N, device=d, dtype=torch.float16) for _ in range(args.pool)]
    tmp = torch.empty(M, K, device=d, dtype=torch.float16)

    def old(i):
        a = ops.layernorm(xs[i], gam, bet, 1e-5, out=tmp) if ln else xs[i]
        ops.gemm(a, wd, outs[i], bias=b, residual=rs[i])

    def new(i):
        rowgemm(xs[i], wp, outs[i], bias=b, residual=rs[i], ln=(gam, bet, 1e-5) if ln else None)

    old(0); new(0)
    ref = outs[0].clone(); old(0)
    err = float((outs[0].float() - ref.float()).norm() / ref.float().norm())      # rowgemm against gemm2 (+ layernorm) on the same operands
    t_old = timeit([lambda i=i: old(i) for i in range(args.pool)])
    t_new = timeit([lambda i=i: new(i) for i in range(args.pool)])
    byt = 2.0 * (M * K + M * N * (2 if res else 1))
    print(f"M {M:6d} N {N:4d} res {int(res)} ln {int(ln)}:  gemm2{' + layernorm' if ln else ''} {t_old:7.1f} us   rowgemm {t_new:7.1f} us   x{t_old / t_new:.2f}   "
          f"{byt / t_new / 1e6:.2f} TB/s algorithmic   rel {err:.1e}", flush=True)

#!/usr/bin/env python3
"""Does the leading dimension of Q / K matter to the forward attention?  The UNet passes column slices of the fused
QKV buffer (ld = 3C); the micro-benchmarks of round 2 used contiguous tensors (ld = C)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import ops  # noqa: E402

dev = "cuda:0"


def timeit(fn, iters=10):
    best = 1e9
    for _ in range(3):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / iters * 1e3)
    return best


for B, H, N, d in ((16, 8, 4096, 40), (16, 8, 1024, 80), (8, 5, 9216, 64)):
    C = H * d
    g = torch.Generator().manual_seed(2)
    qkv = torch.randn(B * N, 3 * C, generator=g).half().to(dev)
    q, k, v = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
    qc, kc = q.contiguous(), k.contiguous()
    vt = ops.transpose(v)
    out = torch.empty(B * N, C, device=dev, dtype=torch.float16)
    lse = torch.empty(B, H, N, device=dev, dtype=torch.float32)
    sc = d ** -0.5
    from sketch2img_amd._lib import lib, check
    def run(Q, K, with_lse):
        check(lib.skg_attn_fwd(Q.data_ptr(), Q.stride(0), K.data_ptr(), K.stride(0), vt.data_ptr(), vt.stride(0), out.data_ptr(),
                               out.stride(0), lse.data_ptr() if with_lse else None, B, H, N, N, N, d, sc,
                               torch.cuda.current_stream().cuda_stream), "attn")
    for name, Q, K, wl in (("q,k slices of qkv (ld 3C), lse", q, k, True), ("q,k slices of qkv (ld 3C)", q, k, False),
                           ("q slice, k contiguous", q, kc, False), ("q,k contiguous (ld C)", qc, kc, False),
                           ("q,k contiguous (ld C), lse", qc, kc, True)):
        t = timeit(lambda: run(Q, K, wl))
        print(f"B{B} H{H} N{N} d{d} {name:34s} {t:8.1f} us {4.0 * B * H * N * N * d / t / 1e6:7.1f} TF/s", flush=True)

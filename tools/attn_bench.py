#!/usr/bin/env python3
"""Micro-benchmark of the attention kernels on the SD1.5 shapes (16 UNet rows forward, 8 rows backward)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import ops  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=10):
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


def case(B, heads, N, dh, Nkv=None, bwd=True):
    C = heads * dh
    Nkv = Nkv or N
    kvs = (Nkv + 7) // 8 * 8
    qkv = torch.randn(B * N, 3 * C, device=DEV).half()
    q = qkv[:, :C]
    if Nkv == N:
        k, v = qkv[:, C:2 * C], qkv[:, 2 * C:]
    else:
        kv = torch.randn(B * kvs, 2 * C, device=DEV).half()
        k, v = kv[:, :C], kv[:, C:]
    scale = dh ** -0.5
    t = timeit(lambda: ops.attn_fwd(q, k, v, B, heads, N, Nkv, kvs, dh, scale, want_lse=True, v_rows=True))
    fl = 4.0 * B * heads * N * Nkv * dh
    print(f"attn fwd  B{B:2d} h{heads} N{N:5d} kv{Nkv:5d} d{dh:3d}: {t * 1e6:8.1f} us  {fl / t / 1e12:6.1f} TF/s", flush=True)
    if not bwd:
        return
    o, lse = ops.attn_fwd(q, k, v, B, heads, N, Nkv, kvs, dh, scale, want_lse=True, v_rows=True)
    do = torch.randn(B * N, C, device=DEV).half()
    delta = ops.attn_bwd_delta(o, do, B, heads, N, dh)
    t = timeit(lambda: ops.attn_bwd_dq(q, k, v, do, lse, delta, B, heads, N, Nkv, kvs, dh, scale))
    print(f"attn dq   B{B:2d} h{heads} N{N:5d} kv{Nkv:5d} d{dh:3d}: {t * 1e6:8.1f} us  {1.5 * fl / t / 1e12:6.1f} TF/s", flush=True)
    if Nkv == N:
        t = timeit(lambda: ops.attn_bwd_dkv(q, k, v, do, lse, delta, B, heads, N, Nkv, dh, scale))
        print(f"attn dkv  B{B:2d} h{heads} N{N:5d} kv{Nkv:5d} d{dh:3d}: {t * 1e6:8.1f} us  {2.0 * fl / t / 1e12:6.1f} TF/s", flush=True)


if __name__ == "__main__":
    case(16, 8, 4096, 40, bwd=False)
    case(8, 8, 4096, 40)
    case(16, 8, 4096, 40, Nkv=77, bwd=False)
    case(16, 8, 1024, 80, bwd=False)
    case(8, 8, 1024, 80)
    case(16, 8, 256, 160, bwd=False)
    case(16, 8, 1024, 80, Nkv=77, bwd=False)
    case(16, 8, 256, 160, Nkv=77, bwd=False)
    case(16, 8, 64, 160, Nkv=77, bwd=False)
    case(8, 8, 256, 160)

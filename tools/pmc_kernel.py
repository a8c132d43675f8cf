#!/usr/bin/env python3
"""One kernel shape, a few launches - the target of `rocprofv3 --pmc ...` passes (tools/pmc_kernels.sh).
    python tools/pmc_kernel.py attn40 | attn64 | conv | conv64 | gemm_short | gemm_ff1 | ffblock | gemm:M:N:K[:res|:geglu] | conv:rows:hw:cin:cout[:res]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import ops  # noqa: E402

dev = "cuda:0"
what = sys.argv[1]
g = torch.Generator().manual_seed(1)
if what.startswith("attn"):
    B, H, N, Nkv, d = (16, 8, 4096, 4096, 40) if what == "attn40" else (8, 5, 9216, 9473, 64)
    kvs = (Nkv + 7) // 8 * 8
    q = torch.randn(B * N, H * d, generator=g).half().to(dev)
    k = torch.randn(B * kvs, H * d, generator=g).half().to(dev)
    v = torch.randn(B * kvs, H * d, generator=g).half().to(dev)
    vt = ops.transpose(v)
    fn = lambda: ops.attn_fwd(q, k, vt, B, H, N, Nkv, kvs, d, d ** -0.5)
elif what == "ffblock":
    from sketch2img_amd.unet import pack_ff_block
    M, C, Fh = 65536, 320, 1280
    x = torch.randn(M, C, generator=g).half().to(dev)
    pack, bias1 = pack_ff_block(torch.randn(2 * Fh, C, generator=g) * C ** -0.5, torch.randn(2 * Fh, generator=g) * 0.1,
                                torch.randn(C, Fh, generator=g) * Fh ** -0.5, dev)
    gam, bet, b2 = torch.ones(C).half().to(dev), torch.zeros(C).half().to(dev), torch.zeros(C).half().to(dev)
    out = torch.empty_like(x)
    fn = lambda: ops.ff_block(x, gam, bet, 1e-5, pack, bias1, b2, out=out)
elif what in ("conv", "conv64"):
    rows, hw, cin, cout = (16, 32, 1920, 640) if what == "conv" else (16, 64, 960, 320)
    x = torch.randn(rows * hw * hw, cin, generator=g).half().to(dev)
    w = (torch.randn(cout, 9 * cin, generator=g) * (9 * cin) ** -0.5).half().to(dev)
    fn = lambda: ops.conv3x3(x, w, rows, hw, hw, 0)
elif what.startswith("conv:"):
    f = what.split(":")
    rows, hw, cin, cout = int(f[1]), int(f[2]), int(f[3]), int(f[4])
    x = torch.randn(rows * hw * hw, cin, generator=g).half().to(dev)
    w = (torch.randn(cout, 9 * cin, generator=g) * (9 * cin) ** -0.5).half().to(dev)
    res = torch.randn(rows * hw * hw, cout, generator=g).half().to(dev) if "res" in f[5:] else None
    out = torch.empty(rows * hw * hw, cout, device=dev, dtype=torch.float16)
    fn = lambda: ops.conv3x3(x, w, rows, hw, hw, 0, out=out, residual=res)
elif what.startswith("gemm:"):
    f = what.split(":")
    M, N, K = int(f[1]), int(f[2]), int(f[3])
    a = torch.randn(M, K, generator=g).half().to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(dev)
    res = torch.randn(M, N, generator=g).half().to(dev) if "res" in f[4:] else None
    out = torch.empty(M, N // 2 if "geglu" in f[4:] else N, device=dev, dtype=torch.float16)
    fn = lambda: ops.gemm(a, w, out=out, residual=res, geglu="geglu" in f[4:])
else:
    M, N, K = (65536, 320, 320) if what == "gemm_short" else (65536, 2560, 320)
    a = torch.randn(M, K, generator=g).half().to(dev)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(dev)
    fn = lambda: ops.gemm(a, w)
ops.gemm(torch.zeros(128, 64, device=dev, dtype=torch.float16), torch.zeros(64, 64, device=dev, dtype=torch.float16))      # (workspace set-up outside the counted launches)
for _ in range(6):
    fn()
torch.cuda.synchronize()

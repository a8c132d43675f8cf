#!/usr/bin/env python3
"""Which K pieces reach the accumulators of the v9 kernel: least-squares fit of its output on the 8-deep partial products."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sketch2img_amd import ops
from sketch2img_amd._lib import lib
dev = "cuda:0"
g = torch.Generator().manual_seed(5)
for M, N, K in [(256, 320, 128), (256, 320, 256), (512, 640, 192)]:
    a = torch.randn(M, K, generator=g).half().to(dev)
    w = torch.randn(N, K, generator=g).half().to(dev)
    out = ops.gemm(a, w).float()
    print("variant", lib.skg_gemm_variant(M, N, K, 0, 0), "M N K", M, N, K)
    P = torch.stack([(a[:, i:i + 8].float() @ w[:, i:i + 8].float().t()).flatten() for i in range(0, K, 8)], 1)      # [M*N, K/8]
    c = torch.linalg.lstsq(P, out.flatten()[:, None]).solution[:, 0]
    print(" piece coefficients:", " ".join(f"{float(x):.2f}" for x in c))
    ref = a.float() @ w.float().t()
    print(" rel err", float((out - ref).norm() / ref.norm()), " residual of the fit", float((P @ c - out.flatten()).norm() / out.norm()))
    # per (32-row block, 32-column block): correlation of out with ref
    for mb in range(0, min(M, 256), 32):
        row = []
        for nb in range(0, 320, 32):
            o, r = out[mb:mb + 32, nb:nb + 32].flatten(), ref[mb:mb + 32, nb:nb + 32].flatten()
            row.append(float((o @ r) / (r @ r)))
        print("  m-block", mb // 32, " ".join(f"{x:5.2f}" for x in row))
M, N, K = 256, 320, 128
a = torch.randn(M, K, generator=g).half().to(dev)
w = torch.randn(N, K, generator=g).half().to(dev)
out = ops.gemm(a, w).float().cpu()
ref = (a.float() @ w.float().t()).cpu()
ok = ((out - ref).abs() < 0.05 * ref.abs().mean()).int()
print("rows 0..3 and 32..35 (m), columns 0..63 (n): 1 = right")
for m in list(range(4)) + list(range(32, 36)) + [64, 200]:
    print(f" m {m:3d} " + "".join(str(int(x)) for x in ok[m, :64]) + " ... " + "".join(str(int(x)) for x in ok[m, 160:192]))
print("fraction right per column (n % 32):", " ".join(f"{float(x):.1f}" for x in ok.float().mean(0)[:32]))
print("fraction right per row (m % 64):", " ".join(f"{float(x):.1f}" for x in ok.float().mean(1)[:64]))
# where do the wrong ones come from? best matching reference column for a wrong entry's column vector
for n in (4, 5, 12, 13, 36):
    col = out[:, n]
    sc = [(float((col - ref[:, k]).abs().mean()), k) for k in range(N)]
    print(f" out column {n} best matches ref column", min(sc)[1], "mean abs diff", round(min(sc)[0], 4))

#!/usr/bin/env python3
"""Round 6 probe: the GEMM step of a Winograd F(2x2, 3x3) convolution IS a split-K launch of the existing kernel on V [M / 4, 16 Cin] x
U [Cout, 16 Cin] (component c = K range [c Cin, (c + 1) Cin), one fp32 slab per component) - what would it run at?  Approximated here
with the shipped split policy (8 slabs of two components + the reduce launch) against the direct implicit-GEMM convolution of the same
layer, config-2 shapes of the 16 x 16 / 8 x 8 levels (16 rows forward, 8 rows backward), cold weights (rotating pool)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import ops

dev = torch.device("cuda:0")
ops.private_buffers      # (import side effects: workspace)
shapes = [(16, 16, 1280, 1280), (16, 16, 2560, 1280), (16, 16, 1920, 1280), (16, 16, 640, 1280), (16, 8, 1280, 1280), (16, 8, 2560, 1280),
          (8, 16, 1280, 1280), (8, 8, 1280, 1280), (8, 16, 1280, 2560), (8, 8, 1280, 2560)]
print(f"{'rows':>4s} {'HxW':>5s} {'Cin':>5s} {'Cout':>5s} | {'direct us':>10s} {'TF/s':>6s} | {'wino GEMM us':>12s} {'TF/s exec':>9s} | {'in + out transform bytes MB':>28s}")
for rows, H, Cin, Cout in shapes:
    M, Mt = rows * H * H, rows * H * H // 4
    npool = max(2, int(400e6 // (Cout * 16 * Cin * 2)))
    X = torch.randn(M, Cin, device=dev).half()
    V = torch.randn(Mt, 16 * Cin, device=dev).half()
    Wd = [torch.randn(Cout, 9 * Cin, device=dev).half() * 0.01 for _ in range(npool)]
    U = [torch.randn(Cout, 16 * Cin, device=dev).half() * 0.01 for _ in range(npool)]
    def t(fn, n=20):
        for i in range(3):
            fn(i)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for i in range(n):
            fn(i)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3
    td = t(lambda i: ops.conv3x3(X, Wd[i % npool], rows, H, H))
    tw = t(lambda i: ops.gemm(V, U[i % npool]))
    fd = 2.0 * M * Cout * 9 * Cin
    fw = 2.0 * Mt * Cout * 16 * Cin
    extra = (M * Cin * 2 + 2 * Mt * 16 * Cin * 2 + 2 * 16 * Mt * Cout * 4 + M * Cout * 2) / 1e6
    print(f"{rows:4d} {H:2d}x{H:<2d} {Cin:5d} {Cout:5d} | {td:10.1f} {fd / td / 1e6:6.0f} | {tw:12.1f} {fw / tw / 1e6:9.0f} | {extra:28.1f}", flush=True)

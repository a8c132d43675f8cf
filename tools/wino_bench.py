#!/usr/bin/env python3
"""Round 6: the Winograd F(2x2, 3x3) convolution (three launches: input transform, split GEMM, output transform) against the implicit GEMM
it replaces, config-2 shapes of the 16 x 16 and 8 x 8 levels (16 rows forward, 8 rows backward), weights rotating through a pool so that
they come from HBM as inside a batch; and conv2 + folded shortcut (one launch) against shortcut GEMM + Winograd conv2 with a residual."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import ops
from sketch2img_amd.unet import pack_conv, pack_conv_wino

dev = torch.device("cuda:0")


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


shapes = [(16, 16, 1280, 1280), (16, 16, 2560, 1280), (16, 16, 1920, 1280), (16, 16, 640, 1280), (8, 16, 1280, 1280), (8, 16, 1280, 2560),
          (8, 16, 1280, 1920), (8, 16, 1280, 640), (16, 8, 1280, 1280), (16, 8, 2560, 1280), (8, 8, 1280, 1280), (8, 24, 1280, 1280)]
print(f"{'rows':>4s} {'HxW':>5s} {'Cin':>5s} {'Cout':>5s} | {'implicit GEMM us':>16s} {'TF/s':>6s} | {'Winograd us':>11s} {'TF/s alg':>8s} {'ratio':>6s}")
for rows, H, Cin, Cout in shapes:
    npool = max(2, int(400e6 // (Cout * 16 * Cin * 2)))
    g = torch.Generator().manual_seed(1)
    X = torch.randn(rows * H * H, Cin, generator=g).half().to(dev)
    ws = [(torch.randn(Cout, Cin, 3, 3, generator=g) * (9 * Cin) ** -0.5) for _ in range(npool)]
    Wd = [pack_conv(w, dev) for w in ws]
    U = [pack_conv_wino(w, dev) for w in ws]
    td = t(lambda i: ops.conv3x3(X, Wd[i % npool], rows, H, H))
    tw = t(lambda i: ops.conv3x3_wino(X, U[i % npool], rows, H, H))
    fl = 2.0 * rows * H * H * Cout * 9 * Cin
    print(f"{rows:4d} {H:2d}x{H:<2d} {Cin:5d} {Cout:5d} | {td:16.1f} {fl / td / 1e6:6.0f} | {tw:11.1f} {fl / tw / 1e6:8.0f} {tw / td:6.2f}", flush=True)
print("\nconv2 (1280 -> 1280) + 1x1 shortcut of the block input (Cx channels): one implicit GEMM with the shortcut folded in (round 5) against shortcut GEMM + Winograd conv2 with a residual")
for rows, H, Cx in [(16, 16, 2560), (16, 16, 1920), (16, 16, 640), (16, 8, 2560)]:
    C = 1280
    g = torch.Generator().manual_seed(2)
    X = torch.randn(rows * H * H, C, generator=g).half().to(dev)
    X2 = torch.randn(rows * H * H, Cx, generator=g).half().to(dev)
    w = torch.randn(C, C, 3, 3, generator=g) * (9 * C) ** -0.5
    wsc = (torch.randn(C, Cx, generator=g) * Cx ** -0.5).half().to(dev)
    wcat = torch.cat([pack_conv(w, dev), wsc], 1).contiguous()
    U = pack_conv_wino(w, dev)
    tf = t(lambda i: ops.conv3x3_sc(X, X2, wcat, rows, H, H))
    def two(i):
        sc = ops.gemm(X2, wsc)
        return ops.conv3x3_wino(X, U, rows, H, H, residual=sc)
    tw = t(two)
    print(f"  rows {rows} {H}x{H} Cx {Cx}: folded {tf:7.1f} us, shortcut GEMM + Winograd {tw:7.1f} us ({tw / tf:.2f})", flush=True)

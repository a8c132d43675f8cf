#!/usr/bin/env python3
"""Bandwidth of the streaming kernels (GroupNorm stats / apply, LayerNorm, GEGLU, axpby, transpose) on the UNet's
largest shapes: bytes moved / time."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import ops  # noqa: E402

D = "cuda:0"


def timeit(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / iters


def line(name, t, nbytes):
    print(f"{name:44s} {t * 1e6:8.1f} us  {nbytes / t / 1e12:6.2f} TB/s", flush=True)


for rows, hw, C in ((16, 4096, 320), (16, 1024, 640), (16, 4096, 960), (8, 4096, 320)):
    M = rows * hw
    x = torch.randn(M, C, device=D).half()
    g, b = torch.ones(C, device=D).half(), torch.zeros(C, device=D).half()
    st = ops.groupnorm_stats(x, rows, hw, 32, 1e-5)
    y = torch.empty_like(x)
    nb = M * C * 2
    line(f"gn_stats  rows {rows} hw {hw} C {C}", timeit(lambda: ops.groupnorm_stats(x, rows, hw, 32, 1e-5)), nb)
    line(f"gn_apply  rows {rows} hw {hw} C {C}", timeit(lambda: ops.groupnorm_apply(x, rows, hw, 32, st, g, b, True, y)), 2 * nb)
    dy = torch.randn(M, C, device=D).half()
    line(f"gn_bwd    rows {rows} hw {hw} C {C}", timeit(lambda: ops.groupnorm_bwd(x, dy, rows, hw, 32, st, g, b, True, out=y)), 5 * nb)
    line(f"ln_fwd    M {M} C {C}", timeit(lambda: ops.layernorm(x, g, b, out=y)), 2 * nb)
    line(f"axpby     M {M} C {C}", timeit(lambda: ops.axpby(x, dy, out=y)), 3 * nb)
M, C = 65536, 320
h = torch.randn(M, 8 * C, device=D).half()
f = torch.empty(M, 4 * C, device=D).half()
line("geglu_fwd M 65536 F 1280", timeit(lambda: ops.geglu(h, out=f, interleaved=True)), M * C * 24)
v = torch.randn(M, C, device=D).half()
line("transpose M 65536 C 320", timeit(lambda: ops.transpose(v)), 2 * M * C * 2)
def gn_fused(x, rows, hw, G, eps, g, b, silu, y):
    """skg_groupnorm_fwd (partial + apply that folds the partials) regardless of the HW threshold in ops.groupnorm"""
    from sketch2img_amd._lib import lib, check
    st = torch.empty(rows, G, 2, device=x.device, dtype=torch.float32)
    check(lib.skg_groupnorm_fwd(x.data_ptr(), x.stride(0), y.data_ptr(), y.stride(0), rows, hw, x.shape[1], G, eps,
                                g.data_ptr(), b.data_ptr(), int(silu), st.data_ptr(),
                                ops._gn_scratch(rows, G, x.device).data_ptr(), torch.cuda.current_stream().cuda_stream), "gn")
    return st


for rows, hw, C in ((16, 4096, 320), (16, 4096, 640), (16, 4096, 960), (8, 4096, 320), (8, 4096, 640), (16, 1024, 640), (16, 256, 1280), (16, 64, 1280)):
    x = torch.randn(rows * hw, C, device=D).half()
    g, b = torch.ones(C, device=D).half(), torch.zeros(C, device=D).half()
    y = torch.empty_like(x)

    def split():
        st = ops.groupnorm_stats(x, rows, hw, 32, 1e-5)
        ops.groupnorm_apply(x, rows, hw, 32, st, g, b, True, y)
    line(f"gn fwd 3 launches rows {rows} hw {hw} C {C}", timeit(split), 3 * x.numel() * 2)
    line(f"gn fwd fused      rows {rows} hw {hw} C {C}", timeit(lambda: gn_fused(x, rows, hw, 32, 1e-5, g, b, True, y)), 3 * x.numel() * 2)

#!/usr/bin/env python3
"""Micro-benchmark of skg_gemm_f16 / skg_conv3x3_f16 on the SD1.5 hot shapes (16 UNet rows = 8 samples).

    python tools/gemm_bench.py            # all shapes, TF/s per shape
    python tools/gemm_bench.py conv64     # one group (used under rocprofv3 --pmc)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import ops  # noqa: E402
from sketch2img_amd._lib import lib  # noqa: E402

DEV = "cuda:0"
ROWS = 16


def timeit(fn, iters):
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


def conv_case(cin, cout, hw, rows=ROWS, mode=0, iters=20):
    x = torch.randn(rows * hw * hw, cin, device=DEV).half()
    w = (torch.randn(cout, 9 * cin, device=DEV) * (9 * cin) ** -0.5).half()
    b = torch.randn(cout, device=DEV).half()
    res = torch.randn(rows * hw * hw, cout, device=DEV).half()
    out = torch.empty(rows * hw * hw, cout, device=DEV, dtype=torch.float16)
    t = timeit(lambda: ops.conv3x3(x, w, rows, hw, hw, mode, out, bias=b, residual=res), iters)
    fl = 2.0 * rows * hw * hw * cout * 9 * cin
    v = lib.skg_gemm_variant(rows * hw * hw, cout, 9 * cin, cin, 1 + mode)
    print(f"conv {cin:5d}->{cout:5d} @{hw:3d}^2 x{rows:2d} rows  variant {v}  {t * 1e6:9.1f} us  {fl / t / 1e12:7.1f} TF/s", flush=True)


def gemm_case(M, N, K, iters=20):
    a = torch.randn(M, K, device=DEV).half()
    w = (torch.randn(N, K, device=DEV) * K ** -0.5).half()
    b = torch.randn(N, device=DEV).half()
    out = torch.empty(M, N, device=DEV, dtype=torch.float16)
    t = timeit(lambda: ops.gemm(a, w, out, bias=b), iters)
    v = lib.skg_gemm_variant(M, N, K, 0, 0)
    print(f"gemm M={M:6d} N={N:5d} K={K:5d}  variant {v}  {t * 1e6:9.1f} us  {2.0 * M * N * K / t / 1e12:7.1f} TF/s", flush=True)


def gemm_res_case(M, N, K, iters=20):
    a = torch.randn(M, K, device=DEV).half()
    w = (torch.randn(N, K, device=DEV) * K ** -0.5).half()
    b = torch.randn(N, device=DEV).half()
    r = torch.randn(M, N, device=DEV).half()
    out = torch.empty(M, N, device=DEV, dtype=torch.float16)
    t = timeit(lambda: ops.gemm(a, w, out, bias=b, residual=r), iters)
    print(f"gemm+res M={M:6d} N={N:5d} K={K:5d}  {t * 1e6:9.1f} us  {2.0 * M * N * K / t / 1e12:7.1f} TF/s", flush=True)


GROUPS = {
    "conv64": lambda: [conv_case(320, 320, 64), conv_case(640, 320, 64), conv_case(960, 320, 64)],
    "conv32": lambda: [conv_case(640, 640, 32), conv_case(1280, 640, 32), conv_case(1920, 640, 32)],
    "conv16": lambda: [conv_case(1280, 1280, 16), conv_case(2560, 1280, 16)],
    "conv8": lambda: [conv_case(1280, 1280, 8), conv_case(2560, 1280, 8), conv_case(1280, 1280, 8, rows=8)],
    "small": lambda: [conv_case(1280, 1280, 16), conv_case(2560, 1280, 16), conv_case(1920, 1280, 16),
                      conv_case(640, 1280, 16), conv_case(1280, 1280, 16, mode=2) if False else None,
                      gemm_case(4096, 1280, 1280), gemm_case(4096, 1280, 5120), gemm_case(4096, 1280, 2560),
                      gemm_case(4096, 3840, 1280), gemm_case(2048, 1280, 1280), gemm_case(8192, 640, 640)],
    "res": lambda: [gemm_res_case(65536, 320, 320), gemm_res_case(65536, 320, 1280), gemm_res_case(16384, 640, 640),
                    gemm_res_case(16384, 640, 2560), gemm_res_case(4096, 1280, 1280), gemm_res_case(4096, 1280, 5120)],
    "gemm": lambda: [gemm_case(65536, 320, 320), gemm_case(65536, 960, 320), gemm_case(65536, 2560, 320),
                     gemm_case(65536, 320, 1280), gemm_case(16384, 640, 640), gemm_case(16384, 5120, 640),
                     gemm_case(16384, 640, 2560), gemm_case(4096, 1280, 1280), gemm_case(4096, 10240, 1280),
                     gemm_case(4096, 1280, 5120), gemm_case(1024, 1280, 1280), gemm_case(131072, 256, 512)],
}

if __name__ == "__main__":
    which = sys.argv[1:] or list(GROUPS)
    for g in which:
        GROUPS[g]()

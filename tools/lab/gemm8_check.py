"""Correctness of the opt-in v8 kernel (256 x 160 tiles, ping-pong schedule; SKG_GEMM8=1), run as a subprocess by
tests/test_gpu_kernels.py::test_gemm8_pingpong_kernel: GEMM and 3x3 convolution shapes it takes (>= 224 tiles), ragged
M, short and long K, bias / alpha / residual / ReLU, against fp32 torch references."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import ops  # noqa: E402
from sketch2img_amd._lib import lib  # noqa: E402

DEV = "cuda:0"
g = torch.Generator().manual_seed(17)
bad = 0
for M, N, K, res, relu, alpha in [(65536, 320, 320, True, False, 1.0), (65536, 320, 1280, True, True, 0.5), (60001, 640, 192, False, False, 1.0), (57345, 320, 128, False, True, 0.5),
                                  (16384, 640, 2560, True, True, 1.0), (8200, 1280, 640, False, False, 0.75)]:
    assert lib.skg_gemm_variant(M, N, K, 0, 0) in (8160, 8320), (M, N, K)
    a = torch.randn(M, K, generator=g).half().to(DEV)
    w = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(DEV)
    b = torch.randn(N, generator=g).half().to(DEV)
    r = torch.randn(M, N, generator=g).half().to(DEV) if res else None
    out = torch.zeros(M, N + 16, device=DEV, dtype=torch.float16)
    ops.gemm(a, w, out=out[:, 8:8 + N], bias=b, residual=r, alpha=alpha, relu=relu)
    ref = alpha * (a.float() @ w.float().t() + b.float())
    if res:
        ref = ref + r.float()
    if relu:
        ref = torch.relu(ref)
    e = float((out[:, 8:8 + N].float() - ref).norm() / ref.norm())
    stray = float(out[:, :8].abs().max() + out[:, 8 + N:].abs().max())
    print(f"gemm8 v{lib.skg_gemm_variant(M, N, K, 0, 0)} M{M} N{N} K{K} res{int(res)} relu{int(relu)}: rel {e:.2e} stray {stray}")
    bad += e > 5e-4 or stray != 0
for rows, hw, cin, cout in [(16, 64, 64, 320), (15, 64, 128, 320), (14, 64, 128, 160), (16, 32, 320, 640)]:
    assert lib.skg_gemm_variant(rows * hw * hw, cout, 9 * cin, cin, 1) in (8160, 8320)
    x = torch.randn(rows, cin, hw, hw, generator=g).half()
    w = (torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5).half()
    b = torch.randn(cout, generator=g).half()
    res = torch.randn(rows * hw * hw, cout, generator=g).half().to(DEV)
    out = ops.conv3x3(x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous().to(DEV),
                      w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous().to(DEV), rows, hw, hw, 0, bias=b.to(DEV), residual=res)
    ref = F.conv2d(x.float().to(DEV), w.float().to(DEV), b.float().to(DEV), padding=1).permute(0, 2, 3, 1).reshape(-1, cout) + res.float()
    e = float((out.float() - ref).norm() / ref.norm())
    print(f"gemm8 conv rows{rows} {cin}->{cout} @{hw}: rel {e:.2e}")
    bad += e > 5e-4
print("ALL OK" if not bad else "FAILED")
sys.exit(1 if bad else 0)

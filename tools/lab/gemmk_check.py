"""Correctness of the withdrawn k-pair kernel (tools/lab/gemmk.hip) - lab build only:
    make -C sketch2img_amd/csrc lab && SKG_LIB=$PWD/sketch2img_amd/libskg_lab.so SKG_GEMMK=1 python -m pytest tools/lab/gemmk_check.py -q
(passed on MI355X in round 3 before the kernel was withdrawn for being slower)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sketch2img_amd import ops as _ops  # noqa: E402


def dev():
    return torch.device("cuda:0")


def nhwc(x):
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def test_gemmk_kpair_kernel(ops=_ops):
    """The k-pair kernel (gemmk.hip: one 8-wave workgroup per 128 x 160 tile, the two wave groups take alternate K tiles and
    exchange their partial sums through LDS) on the launches it is dispatched for - at most 256 tiles - with and without
    the cross-workgroup K slices on top: GEMMs (bias / alpha / residual / ReLU, ragged M, odd K-tile counts, strided
    output views with guard columns) and 3x3 convolutions of the 16x16 / 8x8 levels, against fp32 torch; every launch
    twice, bit-identical (fixed reduction order, no atomics)."""
    from sketch2img_amd._lib import lib
    g = torch.Generator().manual_seed(23)
    for M, N, K, res, relu, alpha in [(4096, 1280, 1280, True, False, 1.0), (4096, 1280, 5120, True, True, 0.5),
                                      (2048, 1280, 1280, False, False, 1.0), (8192, 640, 640, True, False, 1.0),
                                      (1000, 1280, 2560, False, True, 0.75), (512, 1280, 10240, True, False, 1.0),
                                      (4096, 320, 320, False, False, 1.0), (130, 160, 256, True, False, 1.0),
                                      (2048, 640, 4160, False, False, 1.0)]:
        assert lib.skg_gemm_variant(M, N, K, 0, 0) == 9160, (M, N, K)
        a = torch.randn(M, K, generator=g).half().to(dev())
        w = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(dev())
        b = torch.randn(N, generator=g).half().to(dev())
        r = torch.randn(M, N + 8, generator=g).half().to(dev())[:, :N] if res else None
        outs = []
        for _ in range(2):
            out = torch.zeros(M, N + 16, device=dev(), dtype=torch.float16)
            ops.gemm(a, w, out=out[:, 8:8 + N], bias=b, residual=r, alpha=alpha, relu=relu)
            outs.append(out)
        ref = alpha * (a.float() @ w.float().t() + b.float())
        if res:
            ref = ref + r.float()
        if relu:
            ref = torch.relu(ref)
        e = float((outs[0][:, 8:8 + N].float() - ref).norm() / ref.norm())
        stray = float(outs[0][:, :8].abs().max() + outs[0][:, 8 + N:].abs().max())
        print(f"gemmk M{M} N{N} K{K} res{int(res)} relu{int(relu)}: rel {e:.2e} stray {stray}")
        assert e < 5e-4 and stray == 0 and torch.equal(outs[0], outs[1])
    for rows, hw, cin, cout, res in [(16, 16, 1280, 1280, True), (8, 16, 1280, 1280, False), (16, 8, 1280, 1280, True),
                                     (16, 8, 2560, 1280, False), (3, 16, 640, 1280, False), (8, 8, 1280, 640, True),
                                     (2, 32, 320, 320, False)]:
        assert lib.skg_gemm_variant(rows * hw * hw, cout, 9 * cin, cin, 1) == 9160
        x = torch.randn(rows, cin, hw, hw, generator=g).half()
        w = (torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5).half()
        b = torch.randn(cout, generator=g).half()
        r = torch.randn(rows * hw * hw, cout, generator=g).half().to(dev()) if res else None
        xs, ws = nhwc(x).to(dev()), w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous().to(dev())
        o1 = ops.conv3x3(xs, ws, rows, hw, hw, 0, bias=b.to(dev()), residual=r)
        o2 = ops.conv3x3(xs, ws, rows, hw, hw, 0, bias=b.to(dev()), residual=r)
        ref = F.conv2d(x.float().to(dev()), w.float().to(dev()), b.float().to(dev()), padding=1).permute(0, 2, 3, 1).reshape(-1, cout)
        if res:
            ref = ref + r.float()
        e = float((o1.float() - ref).norm() / ref.norm())
        print(f"gemmk conv rows{rows} {cin}->{cout} @{hw}: rel {e:.2e}")
        assert e < 5e-4 and torch.equal(o1, o2)



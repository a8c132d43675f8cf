"""WITHDRAWN EXPERIMENT (round 4; lab build only): the self-finishing split-K launch (gemm2_splitk_kernel, SKG_SPLITK_SELF=1) against
the two-launch path it would have replaced - bit for bit.
    make -C sketch2img_amd/csrc lab && SKG_LIB=$PWD/sketch2img_amd/libskg_lab.so python -m pytest tools/lab/splitk_self_check.py -q -s"""
import os
import sys

import pytest
import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))


def dev():
    return "cuda:0"


def nhwc(x):
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


@pytest.fixture(scope="module")
def ops():
    from sketch2img_amd import ops as o
    return o


def test_split_k_launch_finishes_itself_bitwise_equal_to_the_reduce_kernel(ops, monkeypatch):
    """gemm2_splitk_kernel: the workgroup that draws the last ticket of a tile adds the fp32 slabs in slab order and runs the
    epilogue - the same bits as splitk_reduce_kernel behind the plain launch (default), whichever workgroup
    arrives last: GEMMs and convolutions of the 8 x 8 / 16 x 16 levels (2 ... 8 slices, tiles 64 / 128 / 160 wide), bias +
    residual + ReLU, fp32 output, the (hi, lo) pair epilogue, 40 launches back to back on the same tickets; vs torch."""
    from sketch2img_amd.unet import pack_conv
    g = torch.Generator().manual_seed(77)

    def both(fn, x1, x2):
        """fn(x) through the reduce kernel for two different operands, then the self-finishing launch alternating between them
        (a slab element read stale from an earlier launch - same addresses, other values - would show)"""
        monkeypatch.delenv("SKG_SPLITK_SELF", raising=False)
        a1, a2 = fn(x1), fn(x2)
        monkeypatch.setenv("SKG_SPLITK_SELF", "1")
        ok = True
        for _ in range(12):
            ok = ok and torch.equal(fn(x1), a1) and torch.equal(fn(x2), a2)
        return a1, ok

    for M, N, K, res, relu in [(1024, 1280, 1280, True, False), (512, 1280, 5120, False, True), (2048, 640, 2560, True, False),
                               (300, 320, 4096, False, False), (4096, 1280, 8192, True, False), (128, 64, 1024, False, False)]:
        a_ = torch.randn(M, K, generator=g).half().to(dev())
        a2 = torch.randn(M, K, generator=g).half().to(dev())
        w = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(dev())
        b_ = torch.randn(N, generator=g).half().to(dev())
        r = torch.randn(M, N, generator=g).half().to(dev()) if res else None
        y, same = both(lambda x: ops.gemm(x, w, bias=b_, residual=r, relu=relu), a_, a2)
        ref = a_.float() @ w.float().t() + b_.float()
        ref = (ref + r.float()) if res else ref
        ref = ref.relu() if relu else ref
        e = float((y.float() - ref).norm() / ref.norm())
        print(f"split-K gemm {M}x{N}x{K}: self-finishing == reduce kernel over 24 alternating launches {same}, rel {e:.2e}")
        assert same and e < 5e-4
        y, same = both(lambda x: ops.gemm(x, w, bias=b_, out_f32=True), a_, a2)
        assert same
    for rows, hw, cin, cout in [(16, 8, 1280, 1280), (16, 8, 2560, 1280), (8, 16, 1280, 1280), (2, 16, 640, 1280)]:
        xx = nhwc(torch.randn(rows, cin, hw, hw, generator=g).half()).to(dev())
        x2 = nhwc(torch.randn(rows, cin, hw, hw, generator=g).half()).to(dev())
        w = (torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5).half()
        wp = pack_conv(w, dev())
        r = torch.randn(rows * hw * hw, cout, generator=g).half().to(dev())
        y, same = both(lambda x: ops.conv3x3(x, wp, rows, hw, hw, residual=r), xx, x2)
        ref = F.conv2d(xx.float().view(rows, hw, hw, cin).permute(0, 3, 1, 2), w.float().to(dev()), padding=1).permute(0, 2, 3, 1).reshape(-1, cout) + r.float()
        e = float((y.float() - ref).norm() / ref.norm())
        print(f"split-K conv rows{rows} {cin}->{cout} @{hw}: equal over 24 alternating launches {same}, rel {e:.2e}")
        assert same and e < 5e-4
        lo_a, lo_b = torch.empty_like(r), torch.empty_like(r)      # accuracy mode: pair residual and pair output through the same epilogue
        hi_a, hi_b = torch.empty_like(r), torch.empty_like(r)
        monkeypatch.delenv("SKG_SPLITK_SELF", raising=False)
        ops.conv3x3(xx, wp, rows, hw, hw, out=hi_a, out_lo=lo_a, residual=r, residual_lo=r)
        monkeypatch.setenv("SKG_SPLITK_SELF", "1")
        ops.conv3x3(xx, wp, rows, hw, hw, out=hi_b, out_lo=lo_b, residual=r, residual_lo=r)
        assert torch.equal(hi_a, hi_b) and torch.equal(lo_a, lo_b)

"""Run as a script (subprocess of test_gpu_kernels.py::test_forced_tiles): checks GEMM + every conv gather mode against
torch fp32 with the tile choice forced through SKG_FORCE_BN, so the wide-tile kernels are exercised at sizes that
would normally pick a narrower tile.  Shapes have N % 320 == 0 and ragged M (partial row tiles)."""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import ops  # noqa: E402
from sketch2img_amd.unet import pack_conv, pack_conv_dgrad  # noqa: E402

D = "cuda:0"
TOL = 1.5e-3
fails = []


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half()


def check(name, got, ref):
    r = ((got.float().cpu() - ref).norm() / ref.norm()).item()
    print(f"{name:32s} rel {r:.2e}", flush=True)
    if not r < TOL:
        fails.append(name)


def nhwc(x):
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous()


def from_nhwc(y, B, H, W):
    return y.reshape(B, H, W, -1).permute(0, 3, 1, 2)


# GEMM: ragged M, K = 96 (3 steps: shorter than the pipeline), 160, 1280; bias + residual + alpha; fused GEGLU
for (M, N, K) in [(777, 320, 96), (1000, 640, 160), (513, 320, 1280), (256, 960, 320)]:
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    out = ops.gemm(A.to(D), B.to(D), bias=bias.to(D), residual=res.to(D), alpha=0.5)
    check(f"gemm {M}x{N}x{K}", out, 0.5 * (A.float() @ B.float().t() + bias.float()) + res.float())
M, N, K = 700, 640, 320
A, B, bias = rnd(M, K, seed=5), rnd(N, K, seed=6, scale=K ** -0.5), rnd(N, seed=7)
idx = ops.geglu_interleave_index(N // 2)
y = ops.gemm(A.to(D), B[idx].contiguous().to(D), bias=bias[idx].contiguous().to(D), geglu=True)
h = A.float() @ B.float().t() + bias.float()
check("gemm fused geglu", y, h[:, :N // 2] * F.gelu(h[:, N // 2:]))
o32 = ops.gemm(A.to(D), B.to(D), out_f32=True)
check("gemm f32 out", o32, A.float() @ B.float().t())

# conv: all gather modes, Cout = 320
Bn, Ci, Co, H = 3, 64, 320, 12
x, w, b = rnd(Bn, Ci, H, H, seed=1), rnd(Co, Ci, 3, 3, seed=2, scale=(9 * Ci) ** -0.5), rnd(Co, seed=3)
res = rnd(Bn * H * H, Co, seed=4)
y = ops.conv3x3(nhwc(x).to(D), pack_conv(w, D), Bn, H, H, bias=b.to(D), residual=res.to(D))
check("conv s1", from_nhwc(y.float().cpu(), Bn, H, H),
      F.conv2d(x.float(), w.float(), b.float(), padding=1) + from_nhwc(res.float(), Bn, H, H))
y = ops.conv3x3(nhwc(x).to(D), pack_conv(w, D), Bn, H, H, ops.CONV_S2)
check("conv s2", from_nhwc(y.float().cpu(), Bn, H // 2, H // 2), F.conv2d(x.float(), w.float(), stride=2, padding=1))
y = ops.conv3x3(nhwc(x).to(D), pack_conv(w, D), Bn, H, H, ops.CONV_S2A, bias=b.to(D))
check("conv s2 asym pad", from_nhwc(y.float().cpu(), Bn, H // 2, H // 2),
      F.conv2d(F.pad(x.float(), (0, 1, 0, 1)), w.float(), b.float(), stride=2))
y = ops.conv3x3(nhwc(x).to(D), pack_conv(w, D), Bn, H, H, ops.CONV_UP2)
check("conv up2", from_nhwc(y.float().cpu(), Bn, 2 * H, 2 * H),
      F.conv2d(F.interpolate(x.float(), scale_factor=2.0, mode="nearest"), w.float(), padding=1))
# dgrads produce Cin channels: make Cin = 320
Ci2, Co2 = 320, 64
x2, w2 = rnd(Bn, Ci2, H, H, seed=5), rnd(Co2, Ci2, 3, 3, seed=6, scale=(9 * Ci2) ** -0.5)
gy = rnd(Bn, Co2, H, H, seed=7)
xg = x2.float().requires_grad_(True)
F.conv2d(xg, w2.float(), padding=1).backward(gy.float())
gx = ops.conv3x3(nhwc(gy).to(D), pack_conv_dgrad(w2, D), Bn, H, H, ops.CONV_S1)
check("conv s1 dgrad", from_nhwc(gx.float().cpu(), Bn, H, H), xg.grad)
gy2 = rnd(Bn, Co2, H // 2, H // 2, seed=8)
xg = x2.float().requires_grad_(True)
F.conv2d(xg, w2.float(), stride=2, padding=1).backward(gy2.float())
gx = ops.conv3x3(nhwc(gy2).to(D), pack_conv_dgrad(w2, D), Bn, H // 2, H // 2, ops.CONV_S2T)
check("conv s2 dgrad", from_nhwc(gx.float().cpu(), Bn, H, H), xg.grad)

print("FAILED: " + ", ".join(fails) if fails else "ALL OK")
sys.exit(1 if fails else 0)

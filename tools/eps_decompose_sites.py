#!/usr/bin/env python3
"""Round 6: WHERE do the norm outputs cost the accuracy mode its eps distance?  CPU only (the oracle).  The mode keeps the residual
stream, the conv outputs that feed a norm and the stream-as-operand tensors exact (fp16_storage(skip=("res", "lin_n", "rop")));
what is left (rel ~5e-4) is almost all the fp16 rounding of the GroupNorm(+SiLU) / LayerNorm outputs - the MFMA operands.  This tool
numbers the `_r(., "norm")` rounding points of one evaluation in call order, names them by the module that issued them, and re-runs
the evaluation with ONE GROUP of them exact at a time: the drop in rel^2 is that group's share of the error variance.

    python tools/eps_decompose_sites.py [threads] [t] [seed]
"""
import os
import sys
import traceback

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet as ou

torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else min(32, os.cpu_count() or 1))
T = int(sys.argv[2]) if len(sys.argv) > 2 else 981
SEED = int(sys.argv[3]) if len(sys.argv) > 3 else 7
cfg = ou.SD15
W = ou.init_weights(cfg)
g = torch.Generator().manual_seed(SEED)
x = torch.randn(1, 4, 64, 64, generator=g).half().float()
xx = torch.cat([x, x])
ehs = torch.randn(2, 77, 768, generator=g).half().float()
MODE = ("res", "lin_n", "rop")

_orig_r = ou._r
sites = []          # (index, kind, shape) in call order of one evaluation
exact = set()       # indices of the "norm" / "attn" / "lin_a" rounding points kept exact in this run
counter = [0]
KINDS = ("norm", "attn", "lin_a")


def r(x, kind="lin"):
    if kind in KINDS:
        i = counter[0]
        counter[0] += 1
        if len(sites) <= i:
            sites.append((i, kind, tuple(x.shape)))
        if i in exact:
            return x
    return _orig_r(x, kind)


ou._r = r


def run(on=True):
    counter[0] = 0
    with torch.no_grad(), ou.fp16_storage(on=on, skip=MODE):
        return ou.unet_forward(cfg, W, xx, T, ehs)[0]


ref = run(on=False)
exact.clear()
base = run()
rel0 = float((base - ref).norm() / ref.norm())
print(f"t {T} seed {SEED}: the mode (all {len(sites)} norm / attention / FF-hidden roundings on): eps rel {rel0:.3e} max {float((base - ref).abs().max()):.3e}", flush=True)

# group the rounding points by resolution level walked in order: the spatial size of the tensor tells the level
def level(shape):
    n = shape[-1] * shape[-2] if len(shape) == 4 else shape[1]
    return {4096: 64, 1024: 32, 256: 16, 64: 8}.get(n, n)


# order of one evaluation: down 64, 32, 16, 8, mid 8, up 8, 16, 32, 64, conv_norm_out; split "down" and "up" visits of a level at the
# first change of level after the 8 x 8 visits
groups = {}
seen8 = False
for i, kind, shape in sites:
    lv = level(shape)
    if lv == 8:
        seen8 = True
    side = "up" if (seen8 and lv != 8) else ("mid/8" if lv == 8 else "down")
    groups.setdefault((side, lv, kind), []).append(i)
print(f"{'group kept exact':34s} {'points':>6s} {'eps rel':>10s} {'eps max':>10s} {'share of rel^2':>15s}", flush=True)
for key, idx in groups.items():
    exact.clear(); exact.update(idx)
    e = run()
    rel = float((e - ref).norm() / ref.norm())
    print(f"{str(key):34s} {len(idx):6d} {rel:10.3e} {float((e - ref).abs().max()):10.3e} {1 - (rel / rel0) ** 2:15.3f}", flush=True)
# the last up block, point by point (its errors reach eps with the least averaging)
last = [i for (side, lv, kind), idx in groups.items() if side == "up" and lv == 64 for i in idx]
for i in last:
    exact.clear(); exact.add(i)
    e = run()
    rel = float((e - ref).norm() / ref.norm())
    print(f"point {i:3d} {sites[i][1]:6s} {str(sites[i][2]):24s} {rel:10.3e} {float((e - ref).abs().max()):10.3e} {1 - (rel / rel0) ** 2:15.3f}", flush=True)

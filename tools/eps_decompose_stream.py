#!/usr/bin/env python3
"""Round 6: WHERE does the accuracy mode need its pairs?  CPU only (the oracle).  The mode keeps three classes of tensors exact
(as (hi, lo) pairs): "res" the residual stream, "lin_n" the conv outputs that feed a norm / the residual sum, "rop" the stream where
it is itself a matmul operand.  They cost 8-9 % of the throughput (pair epilogues, norms reading two tensors, K-doubled operands).
This tool numbers those rounding points of one evaluation in call order, groups them by the resolution level they belong to and by
side (down path / 8 x 8 core / up path), and re-runs the evaluation with ONE GROUP ROUNDED to fp16 (everything else as the mode has
it): the growth of rel^2 is what keeping that group as pairs buys.  Groups that buy nothing can run the default fp16 kernels.

    python tools/eps_decompose_stream.py [threads] [t] [seed]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet as ou

torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else min(32, os.cpu_count() or 1))
T = int(sys.argv[2]) if len(sys.argv) > 2 else 981
SEED = int(sys.argv[3]) if len(sys.argv) > 3 else 7
cfg = ou.SD15
W = ou.init_weights(cfg)
g = torch.Generator().manual_seed(SEED)
xx = torch.cat([torch.randn(1, 4, 64, 64, generator=g), torch.randn(1, 4, 64, 64, generator=g)]).half().float()
ehs = torch.randn(2, 77, 768, generator=g).half().float()
MODE = ("res", "lin_n", "rop")

_orig_r = ou._r
sites, rounded, counter = [], set(), [0]


def r(x, kind="lin"):
    if kind in MODE or kind.split("_")[0] in MODE:
        i = counter[0]
        counter[0] += 1
        if len(sites) <= i:
            sites.append((i, kind, tuple(x.shape)))
        return x.half().to(x.dtype) if i in rounded else x
    return _orig_r(x, kind)


ou._r = r


def run(on=True):
    counter[0] = 0
    with torch.no_grad(), ou.fp16_storage(on=on):
        return ou.unet_forward(cfg, W, xx, T, ehs)[0]


ref = run(on=False)
rounded.clear()
base = run()
rel0 = float((base - ref).norm() / ref.norm())
rounded.update(range(len(sites)))
allr = run()
rel1 = float((allr - ref).norm() / ref.norm())
print(f"t {T} seed {SEED}: {len(sites)} res / lin_n / rop points.  All exact (the mode): eps rel {rel0:.3e} max {float((base - ref).abs().max()):.3e};  "
      f"all rounded (default mode): rel {rel1:.3e} max {float((allr - ref).abs().max()):.3e}", flush=True)


def level(shape):
    n = shape[-1] * shape[-2] if len(shape) == 4 else shape[1]
    return {4096: 64, 1024: 32, 256: 16, 64: 8}.get(n, n)


groups, seen8 = {}, False
for i, kind, shape in sites:
    lv = level(shape)
    seen8 = seen8 or lv == 8
    side = "up" if (seen8 and lv != 8) else ("core 8x8" if lv == 8 else "down")
    groups.setdefault((side, lv), []).append(i)
    groups.setdefault((side, lv, kind.split("_")[0] if kind.startswith("rop") else kind), []).append(i)
print(f"{'group ROUNDED to fp16':34s} {'points':>6s} {'eps rel':>10s} {'eps max':>10s} {'rel^2 growth / (default - mode)':>32s}", flush=True)
span = rel1 ** 2 - rel0 ** 2
for key, idx in groups.items():
    rounded.clear(); rounded.update(idx)
    e = run()
    rel = float((e - ref).norm() / ref.norm())
    print(f"{str(key):34s} {len(idx):6d} {rel:10.3e} {float((e - ref).abs().max()):10.3e} {(rel ** 2 - rel0 ** 2) / span:32.3f}", flush=True)
# cumulative: round everything up to and including a point of the network (in evaluation order) - how late can the pairs start?
order = [k for k in groups if len(k) == 2]
acc = []
for key in order:
    acc += groups[key]
    rounded.clear(); rounded.update(acc)
    e = run()
    rel = float((e - ref).norm() / ref.norm())
    print(f"everything through {str(key):15s} {len(acc):6d} {rel:10.3e} {float((e - ref).abs().max()):10.3e} {(rel ** 2 - rel0 ** 2) / span:32.3f}", flush=True)

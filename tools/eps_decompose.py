#!/usr/bin/env python3
"""Which stored tensors cost the fp16 accuracy?  CPU only (the oracle): the full SD1.5 evaluation in fp32, with fp16
storage everywhere, and with one class of stored tensors at a time kept in fp32 (oracle/unet.py fp16_storage(skip=)).
The HIP path itself is compared against the first two in tests/test_gpu_configs.py (it is statistically the same as
the all-fp16-storage evaluation).  Usage: python tools/eps_decompose.py [threads]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet as ou

torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else min(32, os.cpu_count() or 1))
cfg = ou.SD15
W = ou.init_weights(cfg)
g = torch.Generator().manual_seed(7)
x = torch.randn(1, 4, 64, 64, generator=g).half().float()
xx = torch.cat([x, x])
ehs = torch.randn(2, 77, 768, generator=g).half().float()


def run(**kw):
    with torch.no_grad(), ou.fp16_storage(**kw):
        return ou.unet_forward(cfg, W, xx, 981, ehs)[0]


ref = run(on=False)
print(f"{'fp32 kept for':28s} {'eps rel':>10s} {'eps max':>10s}")
for name, skip in (("nothing (all fp16)", ()), ("residual stream", ("res",)), ("norm outputs", ("norm",)),
                   ("conv / linear outputs", ("lin",)), ("q k v and attention out", ("attn",)),
                   ("time embedding path", ("temb",)), ("residual stream + norms", ("res", "norm")),
                   ("all but the residual stream", ("norm", "lin", "attn", "temb"))):
    e = run(on=True, skip=skip)
    print(f"{name:28s} {float((e - ref).norm() / ref.norm()):10.3e} {float((e - ref).abs().max()):10.3e}", flush=True)

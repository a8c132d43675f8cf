#!/usr/bin/env python3
"""Round 5 (cost of the accuracy mode): which of the K-doubled / K-tripled matmul operands of the pair stream
buy how much?  CPU only: the oracle's emulation of the mode, fp16_storage(skip=("res", "lin_n", "rop")), with the residual stream
ROUNDED to fp16 at one class of operand sites at a time (rop_sc: the 14 conv_shortcut GEMMs, rop_dn / rop_up: the 3 + 3 resampling
convolutions, rop_po: the 16 proj_out GEMMs).    python tools/eps_decompose_rop.py [threads] [seed ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet as ou

torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else min(32, os.cpu_count() or 1))
seeds = [int(a) for a in sys.argv[2:]] or [7]
cfg = ou.SD15
W = ou.init_weights(cfg)
ROP = ("rop_sc", "rop_dn", "rop_up", "rop_po")
CASES = [("all four classes as pairs (the mode)", ROP)] + [(f"{k} rounded", tuple(r for r in ROP if r != k)) for k in ROP] + \
        [("rop_up + rop_dn rounded", ("rop_sc", "rop_po")), ("all four rounded", ())]
for seed in seeds:
    g = torch.Generator().manual_seed(seed)
    xx = torch.cat([torch.randn(1, 4, 64, 64, generator=g), torch.randn(1, 4, 64, 64, generator=g)]).half().float()
    ehs = torch.randn(2, 77, 768, generator=g).half().float()
    for t in (981, 21):
        with torch.no_grad():
            ref = ou.unet_forward(cfg, W, xx, t, ehs)[0]
        for name, keep in CASES:
            with torch.no_grad(), ou.fp16_storage(skip=("res", "lin_n") + keep):
                e = ou.unet_forward(cfg, W, xx, t, ehs)[0]
            print(f"seed {seed} t {t:3d}  {name:40s} eps rel {float((e - ref).norm() / ref.norm()):.3e}  max {float((e - ref).abs().max()):.3e}", flush=True)

#!/usr/bin/env python3
"""One guided UNet evaluation of config 0's shape (SD1.5, 1 sample, 32 x 32 latents) twice: with the per-operator cross-attention
launches in the stashing forward (SKG_XATTN_KEEP=0) and with skg_xattn_block_f16_keep; compares eps, taps and every stash entry."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import synthetic, unet as U  # noqa: E402
from sketch2img_amd.config import SD15  # noqa: E402
from sketch2img_amd.unet import HipUNet, Stash  # noqa: E402

dev = "cuda:0"
W = synthetic.unet_state_dict(SD15)
net = HipUNet(SD15, W, dev)
ehs = synthetic.text_embeddings(1)
net.prepare_context(ehs)
h = 32
for seed, t, scale in [(0, 981, 1.0), (1, 881, 1.0), (2, 881, 3.0)]:
    g = torch.Generator().manual_seed(seed)
    x = torch.zeros(2 * h * h, U.CIN_PAD, dtype=torch.float16)
    lat = (torch.randn(h * h, 4, generator=g) * scale).half()
    x[:h * h, :4] = lat
    x[h * h:, :4] = lat
    x = x.to(dev)
    net.prepare_timesteps([t])
    res = []
    for keep in (False, True):
        U._XATTN_KEEP = keep
        st = Stash()
        eps, taps = net.forward(x, t, 2, h, st, want_taps=True, shared_input=True)
        torch.cuda.synchronize()
        res.append((eps.clone(), [a.clone() for a, _ in taps], st))
    (e0, t0, s0), (e1, t1, s1) = res
    rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30))
    print(f"seed {seed} t {t} scale {scale}: eps rel {rel(e1, e0):.2e}; taps " + " ".join(f"{rel(a, b):.1e}" for a, b in zip(t1, t0)))
    for p in s0.tr:
        a, b = s0.tr[p], s1.tr[p]
        if a["H"] != h:
            continue
        M0 = h * h
        cut = lambda d, k, stat=False: d[k] if k in d.get("half", ()) else (d[k][1:] if stat else d[k][M0:])
        msg = [f"p2 {rel(b['p2'], a['p2']):.1e}"]
        for k, stat in (("st2", False), ("q2", False), ("o2", False), ("lse2", True)):
            msg.append(f"{k} {rel(cut(b, k, stat).reshape(-1), cut(a, k, stat).reshape(-1)):.1e}")
        print("   ", p, " ".join(msg))

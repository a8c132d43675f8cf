#!/usr/bin/env python3
"""Runs the bench workload's sampler twice (fewer steps) and compares the latents bit for bit: every kernel on the path
is deterministic by construction (no atomics, fixed reduction orders), so any difference would be a race."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import synthetic
from sketch2img_amd.config import SD15, tap_channels
from sketch2img_amd.lgp import HipLGP
from sketch2img_amd.sampler import DDIMTables, HipSampler
from sketch2img_amd.unet import HipUNet
dev = torch.device("cuda", 0)
S, h, T = 8, 64, int(sys.argv[1]) if len(sys.argv) > 1 else 10
net = HipUNet(SD15, synthetic.unet_state_dict(SD15), dev)
lgp = HipLGP(synthetic.lgp_state_dict(synthetic.lgp_input_dim(SD15)), tap_channels(SD15), dev)
net.prepare_context(synthetic.text_embeddings(S))
tab = DDIMTables.make(T)
net.prepare_timesteps(tab.timesteps.tolist())
lat0, target = synthetic.initial_latents(0, S, h).to(dev), synthetic.sketch_targets(0, S, h).to(dev)
outs = []
for rep in range(3):
    lgp2 = HipLGP(synthetic.lgp_state_dict(synthetic.lgp_input_dim(SD15)), tap_channels(SD15), dev)   # fresh running stats
    x = HipSampler(net, lgp2).sample(lat0, target, T, tables=tab)
    torch.cuda.synchronize()
    outs.append(x.clone())
same = all(torch.equal(outs[0], o) for o in outs[1:])
print(f"{T}-step guided sampling of {S} samples, 3 runs: bitwise identical = {same}; finite = {bool(torch.isfinite(outs[0]).all())}; "
      f"max |diff| = {max(float((outs[0] - o).abs().max()) for o in outs[1:]):.3e}")
sys.exit(0 if same else 1)

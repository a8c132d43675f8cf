#!/usr/bin/env python3
"""Round 5: what do the two correction terms of the accuracy-mode upsampler buy?  The polyphase upsample conv runs on pre-summed
weights (sums of 1 / 2 / 4 fp16 taps: not fp16 numbers), so the mode computes x_hi W_hi + x_lo W_hi + x_hi W_lo (K tripled).  CPU
emulation on the oracle: the mode as built, with the operand rounded (no x_lo term), with the pre-summed weights rounded (no W_lo
term), and with both (= the default-mode launch with a pair output).   python tools/eps_decompose_up.py [threads] [seed ...]"""
import os
import sys

import torch
import torch.nn.functional as F

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet as ou

torch.set_num_threads(int(sys.argv[1]) if len(sys.argv) > 1 else min(32, os.cpu_count() or 1))
seeds = [int(a) for a in sys.argv[2:]] or [7]
cfg = ou.SD15
W = ou.init_weights(cfg)
ROUND_W = False
_interp, _conv = F.interpolate, F.conv2d
G = {0: ([0], [1, 2]), 1: ([0, 1], [2])}      # phase -> tap groups of (dy = 0, dy = 1)


def poly_up(x, w, b):
    """nearest-2x + 3x3 conv (padding 1) as four 2 x 2 convs on the low-res input with pre-summed (optionally fp16-rounded) weights"""
    B, C, H, Wd = x.shape
    out = x.new_zeros(B, w.shape[0], 2 * H, 2 * Wd)
    for a in (0, 1):
        for bb in (0, 1):
            wp = torch.stack([torch.stack([w[:, :, G[a][dy]][:, :, :, G[bb][dx]].sum((2, 3)) for dx in (0, 1)], -1) for dy in (0, 1)], -2)
            if ROUND_W:
                wp = wp.half().float()
            xp = F.pad(x, (1 - bb, bb, 1 - a, a))      # rows i - 1 .. (a = 0) or i .. i + 1 (a = 1)
            out[:, :, a::2, bb::2] = _conv(xp, wp, b)
    return out


class up_patch:
    """inside: the oracle's `interpolate + conv2d` of the three upsamplers runs as poly_up"""
    def __enter__(self):
        self.pending = None
        def interp(h, scale_factor=None, mode=None):
            self.pending = h
            return h
        def conv(h, w, b=None, stride=1, padding=0):
            if self.pending is not None and h is not None and padding == 1 and stride == 1 and w.shape[-1] == 3:
                src, self.pending = self.pending, None
                # (the oracle rounds the interpolated tensor with _r(., "rop_up"): h is that rounded copy of `src` or src itself)
                return poly_up(h if h.shape == src.shape else src, w, b)
            return _conv(h, w, b, stride=stride, padding=padding)
        F.interpolate, F.conv2d = interp, conv
    def __exit__(self, *e):
        F.interpolate, F.conv2d = _interp, _conv


ROP = ("rop_sc", "rop_dn", "rop_po")
for seed in seeds:
    g = torch.Generator().manual_seed(seed)
    xx = torch.cat([torch.randn(1, 4, 64, 64, generator=g), torch.randn(1, 4, 64, 64, generator=g)]).half().float()
    ehs = torch.randn(2, 77, 768, generator=g).half().float()
    for t in (981, 21):
        with torch.no_grad():
            ref = ou.unet_forward(cfg, W, xx, t, ehs)[0]
            with up_patch():
                chk = ou.unet_forward(cfg, W, xx, t, ehs)[0]
        print(f"seed {seed} t {t:3d}  polyphase form == 9-tap form in fp32: rel {float((chk - ref).norm() / ref.norm()):.1e}", flush=True)
        for name, rw, keep in (("x pair, W pair (the mode as built)", False, ROP + ("rop_up",)), ("x rounded, W pair", False, ROP),
                               ("x pair, W rounded", True, ROP + ("rop_up",)), ("x rounded, W rounded (default launch, pair output)", True, ROP),
                               ("... and the downsampler operand rounded too", True, ("rop_sc", "rop_po"))):
            ROUND_W = rw
            with torch.no_grad(), ou.fp16_storage(skip=("res", "lin_n") + keep), up_patch():
                e = ou.unet_forward(cfg, W, xx, t, ehs)[0]
            print(f"seed {seed} t {t:3d}  {name:52s} eps rel {float((e - ref).norm() / ref.norm()):.3e}  max {float((e - ref).abs().max()):.3e}", flush=True)

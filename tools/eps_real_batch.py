#!/usr/bin/env python3
"""Round 6: the accuracy mode's eps distance AT configs[1]'s REAL BATCH (8 samples = 16 rows per evaluation: the instantiations and the
Winograd path the timed region runs), all 8 samples x 3 timesteps = 48 rows against per-sample fp32 oracle evaluations (computed once),
for the candidate settings of unet.HP_PLAIN_LEVELS / HP_NORM_PAIRS / _WINO; and the time of one 16-row evaluation.
    python tools/eps_real_batch.py"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet as ounet
from sketch2img_amd import ops, synthetic, unet as hunet
from sketch2img_amd.config import SD15
from sketch2img_amd.unet import CIN_PAD, HipUNet

DEV = "cuda:0"
torch.set_num_threads(min(32, os.cpu_count() or 1))
cfg = ounet.SD15
W = synthetic.unet_state_dict(SD15)
S, h = 8, 64
ts = (981, 501, 21)
lat = synthetic.initial_latents(0, S, h)
ehs1, ehsS = synthetic.text_embeddings(1), synthetic.text_embeddings(S)
refs = {}
with torch.no_grad():
    for t in ts:
        for si in range(S):
            refs[(t, si)] = ounet.unet_forward(cfg, W, torch.cat([lat[si:si + 1]] * 2), t, ehs1)[0]
        print(f"oracle references t = {t}", flush=True)
L = "up_blocks.3"
N6 = (f"{L}.resnets.1.norm2", f"{L}.resnets.2.norm1", f"{L}.resnets.2.norm2") + tuple(f"{L}.attentions.{j}.norm" for j in range(3))
N9 = tuple(f"{L}.resnets.{j}.{n}" for j in range(3) for n in ("norm1", "norm2")) + tuple(f"{L}.attentions.{j}.norm" for j in range(3))
N8 = tuple(n for n in N9 if n != f"{L}.resnets.0.norm1")
VARIANTS = [("plain 1, 8 sites, Winograd at 8x8 (shipped)", 1, 2, N8, False), ("... + Winograd in the pair zone (16x16)", 1, 2, N8, True),
            ("plain 1, 9 sites + Winograd in the pair zone", 1, 2, N9, True), ("plain 0, 8 sites + Winograd in the pair zone (16x16, 8x8)", 0, 2, N8, True),
            ("plain 2, 6 sites, Winograd", 2, 2, N6, False), ("round 5 (plain 0, no sites, no Winograd)", 0, 0, (), False)]
if "--all" in sys.argv:
    VARIANTS += [("plain 2, 9 sites, Winograd", 2, 2, N9, False), ("plain 2, 6 sites, no Winograd", 2, 0, N6, False), ("plain 1, 6 sites, Winograd at 8x8", 1, 2, N6, False),
                 ("plain 1, 9 sites, Winograd at 8x8", 1, 2, N9, False), ("plain 0, 6 sites", 0, 0, N6, False), ("plain 0, 9 sites", 0, 0, N9, False)]
x16 = ops.nchw_to_nhwc(torch.cat([lat, lat]).to(DEV), CIN_PAD)
print(f"\n{'variant':58s} {'rel mean':>9s} {'rel max':>9s} {'rms abs':>9s} {'max: median':>11s} {'p90 row':>9s} {'WORST':>9s} {'p99.99 |err|':>12s} {'ms / 16-row eval':>17s}")
for name, pl, wino, pairs, hpw in VARIANTS:
    hunet.HP_PLAIN_LEVELS, hunet._WINO, hunet.HP_NORM_PAIRS, hunet._HP_WINO = pl, wino, pairs, hpw
    net = HipUNet(SD15, W, DEV, need_backward=False, residual_fp32=True)
    net.prepare_context(ehsS)
    rels, maxs, errs = [], [], []
    for t in ts:
        e, _ = net.forward(x16, t, 2 * S, h, want_taps=False, shared_input=True)
        g16 = ops.nhwc_to_nchw(e, 2 * S, 4, h, h).cpu()
        for si in range(S):
            got = torch.stack([g16[si], g16[S + si]])
            for row in range(2):
                d = got[row] - refs[(t, si)][row]
                rels.append(float(d.norm() / refs[(t, si)][row].norm())); maxs.append(float(d.abs().max())); errs.append(d.abs().flatten())
    err = torch.cat(errs)
    srt = sorted(maxs)
    p9999 = float(torch.quantile(err[:: max(1, err.numel() // 1000000)], 0.9999))
    for _ in range(2):
        net.forward(x16, 981, 2 * S, h, want_taps=False, shared_input=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        net.forward(x16, 981, 2 * S, h, want_taps=False, shared_input=True)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:58s} {sum(rels) / len(rels):9.3e} {max(rels):9.3e} {float(err.pow(2).mean().sqrt()):9.3e} {srt[len(srt) // 2]:11.3e} {srt[int(0.9 * len(srt))]:9.3e} "
          f"{srt[-1]:9.3e} {p9999:12.3e} {e0.elapsed_time(e1) / 6:17.3f}", flush=True)
    del net
    torch.cuda.empty_cache()

#!/usr/bin/env python3
"""Round 6 (GPU): what does each set of norm-output pairs (unet.HP_NORM_PAIRS) buy the accuracy mode, and what does it cost?
For every variant: HipUNet(residual_fp32=True) with that set -> eps of full-size SD1.5 evaluations (2 rows, 64 x 64 latents) against the
fp32 CPU oracle (references computed once), and the time of one 16-row evaluation (the bench's per-step forward, HIP events).

    python tools/eps_norm_pairs.py [evals=8]        ->  table on stdout (gpurun_out/r06_eps_norm_pairs.txt)
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet as ounet
from sketch2img_amd import ops, synthetic, unet as hunet
from sketch2img_amd.config import SD15
from sketch2img_amd.unet import CIN_PAD, HipUNet

DEV = "cuda:0"
torch.set_num_threads(min(32, os.cpu_count() or 1))
NEV = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cfg = ounet.SD15
W = ounet.init_weights(cfg)
L = "up_blocks.3"
N5 = (f"{L}.resnets.1.norm2", f"{L}.resnets.2.norm2") + tuple(f"{L}.attentions.{j}.norm" for j in range(3))
N9 = tuple(f"{L}.resnets.{j}.{n}" for j in range(3) for n in ("norm1", "norm2")) + tuple(f"{L}.attentions.{j}.norm" for j in range(3))
VARIANTS = [      # (name, deepest levels on the default fp16 kernels, norm outputs kept as pairs)
    ("round 5: pairs everywhere, no norm pairs", 0, ()),
    ("pairs everywhere + 5 norm sites", 0, N5),
    ("8x8 plain + 5 norm sites", 1, N5),
    ("8x8, 16x16 plain + 5 norm sites", 2, N5),
    ("8x8, 16x16 plain + 5 + res.2 norm1", 2, N5 + (f"{L}.resnets.2.norm1",)),
    ("8x8, 16x16 plain + 5 + res.{1,2} norm1", 2, N5 + (f"{L}.resnets.1.norm1", f"{L}.resnets.2.norm1")),
    ("8x8, 16x16 plain + all nine", 2, N9),
    ("8x8, 16x16, 32x32 plain + all nine", 3, N9),
    ("8x8, 16x16 plain, no norm pairs", 2, ()),
    ("DEFAULT mode (every stored tensor fp16)", -1, ()),
]
cases = []
ts = (981, 661, 341, 21)
seed_pairs = ((7, 11), (23, 101), (3, 5), (13, 17))
for i in range(NEV):
    t, seeds = ts[i % 4], seed_pairs[(i // 4) % 4]
    g = torch.Generator().manual_seed(seeds[0] * 1000 + t)
    xx = torch.cat([torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(s)) for s in seeds]).half().float()
    ehs = torch.randn(2, 77, 768, generator=g).half().float()
    with torch.no_grad():
        C, _ = ounet.unet_forward(cfg, W, xx, t, ehs)
    cases.append((xx, ehs, t, C))
    print(f"oracle reference {i + 1} / {NEV} (t = {t}, seeds {seeds})", flush=True)

S = 8
x16 = ops.nchw_to_nhwc(torch.cat([synthetic.initial_latents(0, S, 64)] * 2).to(DEV), CIN_PAD)
ehs16 = synthetic.text_embeddings(S)
print(f"\n{'accuracy-mode variant':44s} {'rel mean':>9s} {'rel max':>9s} {'max mean':>9s} {'max worst':>9s} {'rms':>9s} {'ms / 16-row eval':>17s}")
for name, plv, pairs in VARIANTS:
    hunet.HP_NORM_PAIRS, hunet.HP_PLAIN_LEVELS = pairs, max(plv, 0)
    net = HipUNet(SD15, W, DEV, need_backward=False, residual_fp32=plv >= 0)
    rels, maxs, sq, n = [], [], 0.0, 0
    for xx, ehs, t, C in cases:
        net.prepare_context(ehs)
        e, _ = net.forward(ops.nchw_to_nhwc(xx.to(DEV), CIN_PAD), t, 2, 64)
        A = ops.nhwc_to_nchw(e, 2, 4, 64, 64).cpu()
        for row in range(2):
            d = A[row] - C[row]
            rels.append(float(d.norm() / C[row].norm())); maxs.append(float(d.abs().max()))
            sq += float(d.pow(2).sum()); n += d.numel()
    net.prepare_context(ehs16)
    net.prepare_timesteps([981])
    for _ in range(2):
        net.forward(x16, 981, 2 * S, 64, want_taps=False, shared_input=True)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(6):
        net.forward(x16, 981, 2 * S, 64, want_taps=False, shared_input=True)
    e1.record()
    torch.cuda.synchronize()
    print(f"{name:44s} {sum(rels) / len(rels):9.3e} {max(rels):9.3e} {sum(maxs) / len(maxs):9.3e} {max(maxs):9.3e} {(sq / n) ** 0.5:9.3e} "
          f"{e0.elapsed_time(e1) / 6:17.3f}", flush=True)
    del net
    torch.cuda.empty_cache()

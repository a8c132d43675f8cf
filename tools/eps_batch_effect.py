#!/usr/bin/env python3
"""Round 6: does the accuracy mode's eps distance depend on the BATCH the evaluation runs in?  The same sample (rows 0 and S of the batch)
evaluated (a) alone - 2 rows, the form every earlier full-size parity test used - and (b) inside configs[1]'s real batch of 8 samples (16
rows: other kernel instantiations, the Winograd path of the small maps), with and without the shared CFG prefix, against ONE fp32 oracle
evaluation of that sample.  Variants through the module switches of sketch2img_amd.unet.
    python tools/eps_batch_effect.py [t ...]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import unet as ounet
from sketch2img_amd import ops, synthetic, unet as hunet
from sketch2img_amd.config import SD15
from sketch2img_amd.unet import CIN_PAD, HipUNet

DEV = "cuda:0"
torch.set_num_threads(min(32, os.cpu_count() or 1))
ts = [int(a) for a in sys.argv[1:]] or [981, 21]
cfg = ounet.SD15
W = synthetic.unet_state_dict(SD15)
S, h = 8, 64
lat = synthetic.initial_latents(0, S, h)
ehs1, ehsS = synthetic.text_embeddings(1), synthetic.text_embeddings(S)
samples = (0, 3)
refs = {}
with torch.no_grad():
    for t in ts:
        for si in samples:
            refs[(t, si)] = ounet.unet_forward(cfg, W, torch.cat([lat[si:si + 1]] * 2), t, ehs1)[0]
            print(f"oracle t {t} sample {si}", flush=True)
N6 = hunet.HP_NORM_PAIRS
VARIANTS = [("default build (plain levels 2, Winograd 2)", 2, 2, N6), ("no Winograd", 2, 0, N6), ("pairs everywhere, Winograd", 0, 2, N6),
            ("pairs everywhere, no Winograd", 0, 0, N6), ("round 5 (pairs everywhere, no norm pairs, no Winograd)", 0, 0, ())]
print(f"\n{'variant':58s} {'form':28s} " + " ".join(f"{'t' + str(t) + ' s' + str(si) + ' rel / max':>24s}" for t in ts for si in samples))
for name, pl, wino, pairs in VARIANTS:
    hunet.HP_PLAIN_LEVELS, hunet._WINO, hunet.HP_NORM_PAIRS = pl, wino, pairs
    net = HipUNet(SD15, W, DEV, need_backward=False, residual_fp32=True)
    for form in ("2 rows", "16 rows", "16 rows, shared CFG prefix"):
        cells = []
        for t in ts:
            for si in samples:
                if form == "2 rows":
                    net.prepare_context(ehs1)
                    x = torch.cat([lat[si:si + 1]] * 2)
                    e, _ = net.forward(ops.nchw_to_nhwc(x.to(DEV), CIN_PAD), t, 2, h, want_taps=False)
                    got = ops.nhwc_to_nchw(e, 2, 4, h, h).cpu()
                else:
                    net.prepare_context(ehsS)
                    x = torch.cat([lat, lat])
                    e, _ = net.forward(ops.nchw_to_nhwc(x.to(DEV), CIN_PAD), t, 2 * S, h, want_taps=False, shared_input="shared" in form)
                    g16 = ops.nhwc_to_nchw(e, 2 * S, 4, h, h).cpu()
                    got = torch.stack([g16[si], g16[S + si]])
                d = got - refs[(t, si)]
                cells.append(f"{float(d.norm() / refs[(t, si)].norm()):.3e} / {float(d.abs().max()):.3e}")
        print(f"{name:58s} {form:28s} " + " ".join(f"{c:>24s}" for c in cells), flush=True)
    del net
    torch.cuda.empty_cache()

"""In-process A/B of the fused feed-forward sub-block inside the full SD1.5 UNet (config 2's workload: 8 samples, 64 x 64
latents): the same unguided and guided sampler steps with unet._FF_BLOCK off / on - results compared, steps timed with
HIP events (alternating, so that both see the same box and clocks).

  python tools/ff_block_ab.py [--samples 8] [--reps 6]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from sketch2img_amd import synthetic, unet as unet_mod
from sketch2img_amd.config import SD15, tap_channels
from sketch2img_amd.lgp import HipLGP
from sketch2img_amd.sampler import DDIMTables, HipSampler
from sketch2img_amd.unet import HipUNet


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--samples", type=int, default=8)
    ap.add_argument("--reps", type=int, default=6)
    ap.add_argument("--switch", default="_FF_BLOCK", help="the unet module switch to A/B (_FF_BLOCK, _XATTN_BLOCK, _FF_KEEP)")
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    cfg, h, S, T = SD15, 64, a.samples, 50
    net = HipUNet(cfg, synthetic.unet_state_dict(cfg), dev, need_backward=True)
    lgp = HipLGP(synthetic.lgp_state_dict(synthetic.lgp_input_dim(cfg)), tap_channels(cfg), dev)
    net.prepare_context(synthetic.text_embeddings(S, dim=cfg.cross_attention_dim))
    tab = DDIMTables.make(T)
    net.prepare_timesteps(tab.timesteps.tolist())
    x0 = synthetic.initial_latents(0, S, h).to(dev)
    tgt = synthetic.sketch_targets(0, S, h).to(dev).float().expand_as(x0).contiguous()
    sampler = HipSampler(net, lgp)

    def run(i, on):
        setattr(unet_mod, a.switch, on)
        sampler.reset_history()
        x, _, aux = sampler.step(x0.clone(), x0.clone(), tgt if i <= 0.5 * T else None, tab, i, 7.5, 1.6)
        return x, aux

    for i, name in ((40, "unguided step"), (3, "guided step")):
        res = {}
        for on in (False, True):
            res[on] = run(i, on)
        torch.cuda.synchronize()
        xa, xb = res[False][0].float(), res[True][0].float()
        rel = float((xa - xb).norm() / xa.norm())
        print(f"[{a.switch}] {name}: x_prev fused vs three-launch rel {rel:.3e} max {float((xa - xb).abs().max()):.3e} finite {bool(torch.isfinite(xb).all())}")
        if res[True][1] is not None:
            print("   aux (|grad| stats, loss) three-launch", res[False][1][0].tolist(), "\n   aux fused", res[True][1][0].tolist())
        times = {False: [], True: []}
        for r in range(a.reps):
            for on in (False, True):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                e0.record()
                run(i, on)
                e1.record()
                torch.cuda.synchronize()
                times[on].append(e0.elapsed_time(e1))
        m = {k: sorted(v)[len(v) // 2] for k, v in times.items()}
        print(f"{name}: three launches {m[False]:.3f} ms   fused {m[True]:.3f} ms   ({(m[False] / m[True] - 1) * 100:+.2f} %)   "
              f"all: {['%.2f' % v for v in times[False]]} / {['%.2f' % v for v in times[True]]}", flush=True)


if __name__ == "__main__":
    main()

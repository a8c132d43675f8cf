#!/usr/bin/env python3
"""is the guided-step direction bound of test_sd15_config0_trajectories_vs_oracle (cos > 0.9965 since the
stashing cross-attention launch, 0.9989 before) noise or a regression?  The guided, teacher-forced leg of that test over N
sample indices (initial latents seed 1000 + i, sketch target seed 2000 + i), once per setting of SKG_XATTN_KEEP (read at import:
one subprocess each; the oracle traces are computed once and cached).
    python tools/traj_seeds.py [N=8]  ->  table of 1 - cos per (sample, guided step) and setting, mean / max per setting"""
import os
import pickle
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CACHE = os.path.join(ROOT, "gpurun_out", "traj_oracle_cache.pkl")


def worker(n):
    import torch
    sys.path.insert(0, ROOT)
    from oracle import guidance as og, unet as ounet
    from sketch2img_amd import synthetic
    from sketch2img_amd.config import SD15, tap_channels
    from sketch2img_amd.lgp import HipLGP
    from sketch2img_amd.sampler import DDIMTables, HipSampler
    from sketch2img_amd.unet import HipUNet
    torch.set_num_threads(min(32, os.cpu_count() or 1))      # (every core of a 200-core host: minutes per convolution)
    DEV = "cuda:0"
    cfg = ounet.SD15
    W = synthetic.unet_state_dict(SD15)
    sd = synthetic.lgp_state_dict(synthetic.lgp_input_dim(SD15))
    ehs = synthetic.text_embeddings(1)
    h, T = 32, 10
    cache = pickle.load(open(CACHE, "rb")) if os.path.exists(CACHE) else {}
    net = HipUNet(SD15, W, DEV)
    net.prepare_context(ehs)
    tab = DDIMTables.make(T)
    net.prepare_timesteps(tab.timesteps.tolist())
    sampler = HipSampler(net, HipLGP(sd, tap_channels(SD15), DEV))
    for i in range(n):
        x0, tgt = synthetic.initial_latents(i, 1, h), synthetic.sketch_targets(i, 1, h)
        if i not in cache:
            tr = []
            og.sample_one(cfg, W, dict(sd), ehs, x0, tgt, T, trace=tr)
            cache[i] = [dict(latents=t["latents"], eps=t["eps"],
                             aux=None if t["aux"] is None else dict(alpha=float(t["aux"]["alpha"]), cond_grad=t["aux"]["cond_grad"],
                                                                    loss=float(t["aux"]["loss"]))) for t in tr]
            os.makedirs(os.path.dirname(CACHE), exist_ok=True)
            pickle.dump(cache, open(CACHE, "wb"))
        tr = cache[i]
        noise = x0.to(DEV)
        row = []
        for s in range(T):
            if tr[s]["aux"] is None:
                continue
            x_i = x0 if s == 0 else tr[s - 1]["latents"]
            xp, eps, aux = sampler.step(x_i.to(DEV).contiguous(), noise, tgt.to(DEV), tab, s, 7.5, 1.6, want_eps=True)
            upd_ref = tr[s]["aux"]["alpha"] * tr[s]["aux"]["cond_grad"]
            upd = xp.cpu() - (tr[s]["latents"] - upd_ref)
            cos = float((upd * upd_ref).sum() / (upd.norm() * upd_ref.norm()))
            row.append(1.0 - cos)
        print("ROW", os.environ.get("SKG_XATTN_KEEP", "1"), i, " ".join(f"{v:.3e}" for v in row), flush=True)


if __name__ == "__main__":
    if os.environ.get("SKG_TRAJ_WORKER"):
        worker(int(sys.argv[1]))
    else:
        n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
        rows = {}
        for keep in ("1", "0"):
            r = subprocess.Popen([sys.executable, __file__, str(n)], env=dict(os.environ, SKG_XATTN_KEEP=keep, SKG_TRAJ_WORKER="1"),
                                 stdout=subprocess.PIPE, text=True)
            for ln in r.stdout:
                if ln.startswith("ROW"):
                    print(ln.rstrip(), flush=True)      # (progress: a killed run keeps what it has)
                    f = ln.split()
                    rows.setdefault(f[1], []).append([float(v) for v in f[3:]])
            r.wait()
        print("1 - cos of the guided update direction vs the oracle, teacher-forced, full SD1.5 at config[0]'s shape; rows = sample index, columns = guided steps 0..5")
        for keep in ("1", "0"):
            print(f"SKG_XATTN_KEEP={keep}" + ("  (default: the stashing fused cross-attention launch)" if keep == "1" else "  (per-operator launches in the stashing forward)"))
            allv = []
            for i, row in enumerate(rows.get(keep, [])):
                print(f"  sample {i}: " + " ".join(f"{v:.2e}" for v in row))
                allv += row
            if allv:
                print(f"  mean {sum(allv) / len(allv):.3e}  max {max(allv):.3e}  min cos {1 - max(allv):.5f}")

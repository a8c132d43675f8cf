#!/usr/bin/env python3
"""Round 6 probe (candidate c): two HALF-batches of configs[1] in ONE process on two HIP streams, each replayed from
its captured hipGraphs, phase-shifted - does the latency-bound deep part (16 x 16 / 8 x 8 levels, cond-only backward) of one group hide
under the 64 x 64 / 32 x 32 levels of the other?  Round 2 tried it eager and in lock-step (launch-bound), round 3 as two PROCESSES
(time-sliced): neither is this experiment.  Everything on one box, one process, interleaved:

    A  one sampler, 8 samples, eager (what bench.py times)                       B  the same from captured graphs
    C  two samplers x 4 samples, graphs, two streams, started together           D  ... the second stream delayed by half a step
    E  two samplers x 4 samples, graphs, ONE stream (the price of the half-size launches alone)

    python tools/halfbatch_probe.py [--steps 50] [--mode fast|tolerance] [--reps 3]
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import synthetic
from sketch2img_amd._lib import check, lib
from sketch2img_amd.config import SD15, tap_channels
from sketch2img_amd.lgp import HipLGP
from sketch2img_amd.sampler import DDIMTables, HipSampler
from sketch2img_amd.unet import HipUNet

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=50)
ap.add_argument("--reps", type=int, default=3)
ap.add_argument("--mode", default="fast", choices=("fast", "tolerance"))
args = ap.parse_args()
dev = torch.device("cuda:0")
T, h = args.steps, 64
tab = DDIMTables.make(T)
W = synthetic.unet_state_dict(SD15)
sd_lgp = synthetic.lgp_state_dict(synthetic.lgp_input_dim(SD15))


def make(S, first, graphs):
    net = HipUNet(SD15, W, dev, need_backward=True, residual_fp32=args.mode == "tolerance")
    net.prepare_context(synthetic.text_embeddings(S))
    net.prepare_timesteps(tab.timesteps.tolist())
    smp = HipSampler(net, HipLGP({k: v.clone() for k, v in sd_lgp.items()}, tap_channels(SD15), dev), use_graphs=graphs)
    return smp, synthetic.initial_latents(first, S, h).to(dev), synthetic.sketch_targets(first, S, h).to(dev)


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    return min(ts), sum(ts) / len(ts)


res = {}
# A / B: one sampler, 8 samples
s8, l8, t8 = make(8, 0, False)
res["A  1 x 8 eager"] = timed(lambda: s8.sample(l8, t8, T, tables=tab), args.reps)
s8.use_graphs = True
res["B  1 x 8 graphs"] = timed(lambda: s8.sample(l8, t8, T, tables=tab), args.reps)
ref8 = s8.last_latents.clone()
del s8
torch.cuda.empty_cache()

# C / D / E: two samplers x 4 samples
sa, la, ta = make(4, 0, True)
sb, lb, tb_ = make(4, 4, True)
sa.sample(la, ta, T, tables=tab)          # capture
sb.sample(lb, tb_, T, tables=tab)
torch.cuda.synchronize()
ea, eb = list(sa._graphs.values())[0], list(sb._graphs.values())[0]
st_a, st_b = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
probe_out = torch.empty(512 * 512, device=dev, dtype=torch.float32)


def reset(ent, lat, tgt):
    ent["xs"].copy_(lat); ent["ns"].copy_(lat); ent["ts"].copy_(tgt.expand_as(lat)); ent["x0b"].zero_()


def two_streams(delay_iters):
    reset(ea, la, ta); reset(eb, lb, tb_)
    torch.cuda.synchronize()
    if delay_iters:
        with torch.cuda.stream(st_b):
            check(lib.skg_box_probe_mfma(probe_out.data_ptr(), delay_iters, st_b.cuda_stream), "delay")
    for i in range(T):
        with torch.cuda.stream(st_a):
            ea["graphs"][i].replay()
        with torch.cuda.stream(st_b):
            eb["graphs"][i].replay()


def one_stream():
    reset(ea, la, ta); reset(eb, lb, tb_)
    for i in range(T):
        ea["graphs"][i].replay()
        eb["graphs"][i].replay()


res["E  2 x 4 graphs, one stream"] = timed(one_stream, args.reps)
res["C  2 x 4 graphs, two streams, together"] = timed(lambda: two_streams(0), args.reps)
xa, xb = ea["xs"].clone(), eb["xs"].clone()
# half a step of one group: ~ (E / 2) / T / 2 seconds; the delay kernel runs ~0.67 us per iteration at 2 PFLOP/s (512 workgroups: 1.34 us)
half_step = res["E  2 x 4 graphs, one stream"][0] / 2 / T / 2
iters = max(1, int(half_step / 1.34e-6))
res[f"D  2 x 4 graphs, two streams, second delayed by {half_step * 1e3:.1f} ms"] = timed(lambda: two_streams(iters), args.reps)
same = torch.equal(ea["xs"], xa) and torch.equal(eb["xs"], xb)
fin = bool(torch.isfinite(ea["xs"]).all() and torch.isfinite(eb["xs"]).all())
print(f"configs[1] sampling loop without the VAE decode, {T} DDIM steps, mode {args.mode}; best / mean of {args.reps} runs after one warm-up")
for k, (best, mean) in res.items():
    print(f"  {k:62s} {best * 1e3:9.1f} ms  {mean * 1e3:9.1f} ms   {8 / best:6.3f} images/s")
print(f"  two-stream results finite {fin}, bit-equal between the undelayed and the delayed schedule {same}; "
      f"4-sample graphs vs the 8-sample batch, samples 0..3 rel {float((xa - ref8[:4]).norm() / ref8[:4].norm()):.2e}")

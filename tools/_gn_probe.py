"""PROBE (not shipped): run bench.one_batch with the GroupNorm statistics passes of the UNet replaced by cached
statistics from an identical earlier batch - the upper bound of what producer-side statistics could save."""
import os, sys, time, json, types
sys.path.insert(0, "/root/repo")
os.chdir("/root/repo")
import torch
import bench
from sketch2img_amd import ops

state = dict(i=0, cache={}, on=False)
_gn = ops.groupnorm


def groupnorm(X, rows, HW, groups, eps, gamma, beta, silu, out=None):
    if HW < 1024 or HW > 4096:
        return _gn(X, rows, HW, groups, eps, gamma, beta, silu, out)
    k = state["i"]; state["i"] += 1
    if state["on"] and k in state["cache"]:
        st = state["cache"][k]
        return ops.groupnorm_apply(X, rows, HW, groups, st, gamma, beta, silu, out), st
    y, st = _gn(X, rows, HW, groups, eps, gamma, beta, silu, out)
    state["cache"][k] = st
    return y, st


ops.groupnorm = groupnorm
sys.argv = ["bench.py", "--no-cpu-baseline", "--no-roofline"]
args = bench.parse()
dev = torch.device("cuda:0")
w = bench.build_workload(args, 0, 1, dev, None)
for mode in (False, True, False, True):
    state["on"] = mode
    state["i"] = 0; w["one_batch"](); torch.cuda.synchronize()
    t0 = time.time()
    for _ in range(2):
        state["i"] = 0; x = w["one_batch"]()
    torch.cuda.synchronize()
    dt = (time.time() - t0) / 2
    print(f"cached GN stats={mode}: {dt * 1e3:.1f} ms per batch, {w['S'] / dt:.3f} img/s, finite={bool(torch.isfinite(x.float()).all())}, GN calls {state['i']}", flush=True)

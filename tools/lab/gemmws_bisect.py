#!/usr/bin/env python3
"""Which launch of a full SD1.5 evaluation differs between the weight-stationary kernel and the tile kernel?
Wraps ops.gemm: every N = K = 320 launch the streaming kernel takes is repeated with SKG_GEMMWS=0 and compared."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sketch2img_amd import ops, synthetic
from sketch2img_amd._lib import lib
from sketch2img_amd.config import SD15
from sketch2img_amd.unet import CIN_PAD, HipUNet

DEV = "cuda:0"
real = ops.gemm
bad = []

def gemm(A, B, out=None, **k):
    M, K = A.shape
    N = B.shape[0]
    os.environ["SKG_GEMMWS"] = "1"
    if not (N == 320 and K == 320 and M >= 32768) or k.get("gn_stats") or k.get("geglu") or k.get("out_f32") or k.get("relu"):
        return real(A, B, out, **k)
    res = k.get("residual")
    alias = out is not None and res is not None and out.data_ptr() == res.data_ptr()
    res_copy = res.clone() if res is not None else None
    k2 = dict(k); k2["residual"] = res_copy
    os.environ["SKG_GEMMWS"] = "0"
    ref = real(A, B, None, **k2)
    os.environ["SKG_GEMMWS"] = "1"
    got = real(A, B, out, **k)
    same = torch.equal(ref, got)
    fin = bool(torch.isfinite(got).all())
    info = (f"M{M} lda{A.stride(0)} ldb{B.stride(0)} ldc{got.stride(0)} bias={k.get('bias') is not None} res={res is not None}"
            f" ldr={res.stride(0) if res is not None else 0} alias={alias} alpha={k.get('alpha', 1.0)} A%16={A.data_ptr() % 16} "
            f"same={same} finite={fin}")
    print(info, flush=True)
    if not same:
        d = (ref.float() - got.float()).abs()
        rows = torch.nonzero(d.amax(1) > 0).flatten()
        cols = torch.nonzero(d.amax(0) > 0).flatten()
        print("   differing rows", rows[:8].tolist(), "...", rows[-4:].tolist(), "count", rows.numel(), "cols", cols[:8].tolist(), "count", cols.numel())
        bad.append(info)
    return got

ops.gemm = gemm
net = HipUNet(SD15, synthetic.unet_state_dict(SD15), DEV, need_backward=True)
rows = 16
net.prepare_context(synthetic.text_embeddings(rows // 2))
xb = synthetic.initial_latents(0, rows // 2, 64)
x32 = ops.nchw_to_nhwc(torch.cat([xb, xb]).to(DEV), CIN_PAD)
from sketch2img_amd.unet import Stash
st = Stash()
eps, taps = net.forward(x32, 981, rows, 64, st, shared_input=True)
torch.cuda.synchronize()
print("eps finite", bool(torch.isfinite(eps).all()), "bad launches", len(bad))

# ---- backward-to-input of the same evaluation (cond rows only: M = 32768 launches) --------------------------------------
from sketch2img_amd.config import tap_channels
from sketch2img_amd.lgp import HipLGP
from sketch2img_amd.sampler import DDIMTables, HipSampler
lgp = HipLGP(synthetic.lgp_state_dict(sum(tap_channels(SD15)) + 40), tap_channels(SD15), DEV)
keep = {}
S = rows // 2
noise = xb.to(DEV).float()
out = lgp.forward(taps, noise, 0.5, S, 64, keep)
tg, loss = lgp.backward(out, synthetic.sketch_targets(0, S, 64).to(DEV).float(), keep)
grad = net.backward(st, tg)
torch.cuda.synchronize()
print("grad finite", bool(torch.isfinite(grad).all()), "bad launches", len(bad))
# ---- whole guided steps, in line and forked ----------------------------------------------------------------------------
ops.gemm = real
tab = DDIMTables.make(50)
for ws in ("0", "1"):
    for fork in (False, True):
        os.environ["SKG_GEMMWS"] = ws
        smp = HipSampler(net, lgp)
        smp.fork_guidance = fork
        x = xb.to(DEV).float()
        fin = []
        for i in range(3):
            x, e, aux = smp.step(x, noise, synthetic.sketch_targets(0, S, 64).to(DEV).float(), tab, i, 7.5, 1.6, want_eps=True)
            torch.cuda.synchronize()
            fin.append((bool(torch.isfinite(x).all()), bool(torch.isfinite(e).all()), bool(torch.isfinite(aux).all())))
        print(f"SKG_GEMMWS={ws} fork={fork}: finite (x, eps, aux) per step {fin}  |x| {float(x.abs().max()):.3f}", flush=True)

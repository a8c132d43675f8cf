#!/usr/bin/env python3
"""How much of the 2-3 % direction difference between the guidance update of this repo and the reference's own
apply_anti_gradient is the ReLU-gate floor, and how much is OURS (the moved fp16 rounding point of the re-associated
layer 0)?   VERDICT r2 weak #1 / next #7.

Runs ONLY in the build container (imports /root/reference under the names-only diffusers stub of tools/gen_golden.py; CPU).
For several seeds it builds the golden-vector fixture of tools/gen_golden.py (nine differentiable toy taps, fp16-valued
like the real UNet taps) and computes the guidance direction g = -d loss / d x_in (cond chunk) five ways:

    R   the reference: modules/pipeline.py:141-161 with the fp16 LatentEdgePredictor on CPU (its own autograd)
    A   oracle as written: resize -> concat -> fp16 cast -> Linear 0 ...   (oracle/lgp.py, rounding where R rounds)
    B   oracle with layer 0 RE-ASSOCIATED exactly as lgp.hip / sketch2img_amd/lgp.py do it: every tap times its slice of
        W0 at NATIVE resolution (fp32), the 512-channel partial sums resized in fp32, the 40 noise-level / sinusoid
        channels rounded to fp16 and multiplied separately, one fp16 rounding of the sum
    Bg  arm B with every ReLU gate FORCED to arm A's gates (what is left is the rounding-point shift without gate flips)
    D   the smooth fp64 gradient (no rounding anywhere)

and reports relative distances / cosines between them plus the fraction of ReLU gates that differ.  If B is measurably
further from R than A is, part of the 2-3 % is ours; if B ~ A (and Bg ~ A to << 1 %), it is the gate floor.

    python tools/lgp_rounding_floor.py [--seeds 8] [--h 16]
"""
import argparse
import math
import os
import sys
import types

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))

from gen_golden import TAP_C, import_reference, seeded_lgp, tap_sizes  # noqa: E402

# the reference's `modules` is a namespace package (no __init__.py) and this repo's alias package `modules/` is a regular
# one, which would win whatever the path order: import the reference BEFORE the repo root goes on sys.path
REFERENCE = import_reference()
sys.path.insert(0, ROOT)
from oracle import lgp as olgp  # noqa: E402


def r16(x):
    return x + (x.detach().half().to(x.dtype) - x.detach())


def lgp_from(z0, sd, dt, emulate, gates=None, record=None):
    """Layers 0's output z0 (pre-ReLU, already rounded as the arm wants) -> LGP output; train-mode BatchNorm."""
    z = z0
    for i in range(4):
        if record is not None:
            record.append((z > 0).detach())
        z = z * gates[i].to(z.dtype) if gates is not None else torch.relu(z)
        bn = olgp.BNS[i]
        mean, var = z.mean(0), z.var(0, unbiased=False)
        z = (z - mean) * torch.rsqrt(var + 1e-5) * sd[f"layers.{bn}.weight"].to(dt) + sd[f"layers.{bn}.bias"].to(dt)
        z = r16(z) if emulate else z
        z = z @ sd[f"layers.{olgp.LIN[i + 1]}.weight"].to(dt).t() + sd[f"layers.{olgp.LIN[i + 1]}.bias"].to(dt)
        z = r16(z) if emulate else z
    return z


def direction(out, x_in, target, h):
    o = out.reshape(2, h, h, -1).permute(0, 3, 2, 1)
    loss = F.mse_loss(target.to(o.dtype), o.chunk(2)[1], reduction="mean")
    return (-torch.autograd.grad(loss, x_in)[0]).chunk(2)[1].detach().double(), float(loss)


def rel(a, b):
    return float((a - b).norm() / b.norm())


def cos(a, b):
    return float((a * b).sum() / (a.norm() * b.norm()))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seeds", type=int, default=8)
    ap.add_argument("--h", type=int, default=16)
    args = ap.parse_args()
    torch.set_num_threads(4)
    LatentEdgePredictor, _, AntiGradientPipeline = REFERENCE
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    acp = torch.cumprod(1 - betas, 0)
    h, tstep = args.h, 501
    E = sum(TAP_C)
    rows = []
    for seed in range(args.seeds):
        lgp = seeded_lgp(LatentEdgePredictor, E + 40, seed=900 + seed)
        lgp.train()
        sd = {k: (v.float() if v.dtype.is_floating_point else v) for k, v in lgp.state_dict().items()}
        g = torch.Generator().manual_seed(50 + seed)
        convs = [0.5 * torch.randn(c, 4, 1, 1, generator=g) for c in TAP_C]
        x = torch.randn(1, 4, h, h, generator=g)
        latents = x + 0.1 * torch.randn(1, 4, h, h, generator=g)
        noise = torch.randn(1, 4, h, h, generator=g)
        target = 0.18215 * torch.randn(1, 4, h, h, generator=g)

        def taps_of(x_in, dt):
            out = []
            for w, s in zip(convs, tap_sizes(h)):
                f = F.adaptive_avg_pool2d(torch.tanh(F.conv2d(x_in.to(dt), w.to(dt))), s)
                out.append(r16(f))                       # the UNet's taps are fp16 tensors (.float() of fp16 outputs)
            return out

        # ---- R: the reference --------------------------------------------------------------------------------------
        p = AntiGradientPipeline.__new__(AntiGradientPipeline)
        p.scheduler = types.SimpleNamespace(alphas_cumprod=acp)
        p.lgp_model = lgp
        gatesR = []
        hooks = [m.register_forward_hook(lambda mod, i, o, st=gatesR: st.append((o > 0).detach()))
                 for m in lgp.layers if isinstance(m, torch.nn.ReLU)]
        x_in = torch.cat([x] * 2).requires_grad_(True)
        with torch.enable_grad():
            p.feature_blocks = [types.SimpleNamespace(output=f.float()) for f in taps_of(x_in, torch.float32)]
            outR = p.apply_anti_gradient(x_in, latents, noise, torch.tensor(tstep), target, 1.6)
        for hk in hooks:
            hk.remove()
        gR = (outR.detach() - latents).double()          # alpha * g: direction only is compared
        gR = gR / gR.norm()

        nl = ((1 - acp[tstep]) ** 0.5).reshape(1, 1, 1, 1) * noise
        t2 = torch.cat([nl] * 2)

        def arm(kind, dt=torch.float32, gates=None, record=None):
            emulate = kind != "D"
            x_in = torch.cat([x] * 2).to(dt).requires_grad_(True)
            with torch.enable_grad():
                taps = taps_of(x_in, dt)
                pos = olgp.positional_channels(t2.to(dt), 9)
                extra = torch.cat((t2.to(dt), pos), 1)                          # (2, 40, h, h)
                W0, b0 = sd["layers.0.weight"].to(dt), sd["layers.0.bias"].to(dt)
                flat = lambda z: z.permute(0, 3, 2, 1).reshape(-1, z.shape[1])   # "(b w h) c"
                if kind in ("A", "D"):
                    feats = torch.cat([F.interpolate(tp, size=h, mode="bilinear") for tp in taps] + [extra], 1)
                    z = flat(feats)
                    z = r16(z) if emulate else z
                    z0 = z @ W0.t() + b0
                else:                                   # B: per-tap GEMM at native resolution, fp32 partial sums resized
                    acc, off = 0, 0
                    for tp, c in zip(taps, TAP_C):
                        pp = torch.einsum("bchw,oc->bohw", tp, W0[:, off:off + c])
                        acc = acc + F.interpolate(pp, size=h, mode="bilinear")
                        off += c
                    acc = acc + torch.einsum("bchw,oc->bohw", r16(extra), W0[:, E:])
                    z0 = flat(acc) + b0
                z0 = r16(z0) if emulate else z0
                out = lgp_from(z0, sd, dt, emulate, gates, record)
                return direction(out, x_in, target, h)

        gatesA, gatesB = [], []
        gA, lA = arm("A", record=gatesA)
        gB, lB = arm("B", record=gatesB)
        gBg, _ = arm("B", gates=gatesA)
        gD, lD = arm("D", dt=torch.float64)
        n = lambda v: v / v.norm()
        gA, gB, gBg, gD = n(gA), n(gB), n(gBg), n(gD)
        flipAB = sum(int((a != b).sum()) for a, b in zip(gatesA, gatesB)) / sum(a.numel() for a in gatesA)
        flipAR = sum(int((a != b).sum()) for a, b in zip(gatesA, gatesR)) / sum(a.numel() for a in gatesA)
        flipBR = sum(int((a != b).sum()) for a, b in zip(gatesB, gatesR)) / sum(a.numel() for a in gatesA)
        rows.append(dict(seed=seed, AR=rel(gA, gR), BR=rel(gB, gR), AB=rel(gA, gB), BgA=rel(gBg, gA), DR=rel(gD, gR),
                         AD=rel(gA, gD), BD=rel(gB, gD), cosAR=cos(gA, gR), cosBR=cos(gB, gR),
                         flipAB=flipAB, flipAR=flipAR, flipBR=flipBR))
        r = rows[-1]
        print(f"seed {seed}: |A-R| {r['AR']:.4f}  |B-R| {r['BR']:.4f}  |A-B| {r['AB']:.4f}  |Bg-A| {r['BgA']:.5f}  |D-R| {r['DR']:.4f}  "
              f"|A-D| {r['AD']:.4f}  |B-D| {r['BD']:.4f}  cos(A,R) {r['cosAR']:.5f}  cos(B,R) {r['cosBR']:.5f}  "
              f"gates differing A/B {100 * flipAB:.3f} %  A/R {100 * flipAR:.3f} %  B/R {100 * flipBR:.3f} %", flush=True)
    mean = lambda k: sum(r[k] for r in rows) / len(rows)
    mx = lambda k: max(r[k] for r in rows)
    print(f"\nh = {h}, {len(rows)} seeds, unit-norm directions, relative Frobenius distance (mean / max):")
    for k, what in [("AR", "oracle as written        vs reference"), ("BR", "re-associated layer 0    vs reference"),
                    ("AB", "re-associated            vs as written"), ("BgA", "re-associated, A's gates vs as written"),
                    ("DR", "fp64 smooth gradient     vs reference"), ("AD", "oracle as written        vs fp64"),
                    ("BD", "re-associated            vs fp64")]:
        print(f"  {what}: {mean(k):.4f} / {mx(k):.4f}")
    print(f"  ReLU gates differing: A vs B {100 * mean('flipAB'):.3f} %, A vs R {100 * mean('flipAR'):.3f} %, B vs R {100 * mean('flipBR'):.3f} %")
    print(f"  min cos: A,R {min(r['cosAR'] for r in rows):.5f}   B,R {min(r['cosBR'] for r in rows):.5f}")


if __name__ == "__main__":
    main()

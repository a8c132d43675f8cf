"""GPU parity tests (-m gpu) of the assembled hot path - LGP, UNet forward, UNet backward-to-input, the
guidance step and the sampling loop - against (a) the golden vectors made from the reference's own code
and (b) the CPU oracle on identical seeded inputs.  Everything goes through the C ABI (libskg.so)."""
import math
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from tests.util import load_npz, report, sd_from_npz

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def nhwc16(x):
    return x.permute(0, 2, 3, 1).reshape(-1, x.shape[1]).contiguous().half().to(DEV)


def from_nhwc(y, B, H, W):
    return y.float().cpu().reshape(B, H, W, -1).permute(0, 3, 1, 2)


def tap_sizes(h):
    return [h // 2, h // 4, h // 8, h // 8, h // 8, h // 8, h // 4, h // 2, h]


# ------------------------------------------------------------------------------------------------------ LGP
@pytest.mark.parametrize("h", [8, 16])
@pytest.mark.parametrize("training", [True, False])
def test_lgp_forward_matches_reference_golden(h, training):
    """HIP LGP vs outputs of the reference's LatentEdgePredictor (fp16 module, CPU)."""
    from sketch2img_amd.lgp import HipLGP
    d = load_npz(f"lgp_fwd_h{h}.npz")
    sd = sd_from_npz(d)
    x, t = torch.from_numpy(d["x"]), torch.from_numpy(d["t"])
    lgp = HipLGP(sd, [x.shape[1]], DEV, training=training)
    out = lgp.forward([(nhwc16(x), h)], t[:1].contiguous().to(DEV), 1.0, 1, h)
    ref = torch.from_numpy(d["y_train" if training else "y_eval"]).reshape(2, h, h, 4).permute(0, 2, 1, 3)  # (b w h)->(b y x)
    got = out[:, :4].float().cpu().reshape(2, h, h, 4)
    _, m = report(f"lgp fwd golden h{h} train={training}", got, ref)
    assert m <= 4 * 2 ** -8                      # a few fp16 ulps at |y| <= 8 (same bound as the oracle's)
    assert out[:, 4:].abs().max() == 0
    if training:
        for l, i in enumerate((2, 5, 8, 11)):
            for nm, got_r in (("running_mean", lgp.running_mean[l]), ("running_var", lgp.running_var[l])):
                ref_r = torch.from_numpy(d[f"sd_after.layers.{i}.{nm}"]).float()
                assert (got_r.cpu() - ref_r).abs().max() <= 1e-3 * max(1.0, float(ref_r.abs().max()))
        assert lgp.num_batches_tracked == [1, 1, 1, 1]


@pytest.mark.parametrize("h", [8, 16])
def test_guidance_step_matches_reference_golden(h):
    """Reference apply_anti_gradient (toy differentiable feature extractor on the CPU side) vs HIP LGP
    forward + backward-to-features + guidance update."""
    from sketch2img_amd import ops
    from sketch2img_amd.lgp import HipLGP
    from sketch2img_amd.sampler import DDIMTables
    d = load_npz(f"guidance_h{h}.npz")
    sd = sd_from_npz(d)
    x = torch.from_numpy(d["x"])
    x_in = torch.cat([x] * 2).requires_grad_(True)
    sizes = tap_sizes(h)
    taps = [F.adaptive_avg_pool2d(torch.tanh(F.conv2d(x_in, torch.from_numpy(d[f"conv{i}"]))), s)
            for i, s in enumerate(sizes)]
    lgp = HipLGP(sd, [t.shape[1] for t in taps], DEV)
    tab = DDIMTables.make(50)
    assert np.array_equal(tab.alphas_cumprod.numpy(), d["alphas_cumprod"])      # bit-exact fp32 table
    keep = {}
    noise = torch.from_numpy(d["noise"]).to(DEV)
    out = lgp.forward([(nhwc16(t.detach()), s) for t, s in zip(taps, sizes)], noise, tab.sigma(int(d["t"])), 1, h, keep)
    grads, loss = lgp.backward(out, torch.from_numpy(d["target"]).to(DEV), keep)
    # push the HIP feature gradients (cond row) back through the toy extractor on the CPU
    tot = 0
    for t, s, g in zip(taps, sizes, grads):
        gc = from_nhwc(g, 1, s, s)
        tot = tot + (t[1:2] * gc).sum()
    grad_x = torch.autograd.grad(tot, x_in)[0][1:2]                # d loss / d x_in, cond row (x LOSS_SCALE)
    gn = torch.zeros(h * h, 8, dtype=torch.float16)
    gn[:, :4] = grad_x.permute(0, 2, 3, 1).reshape(h * h, 4).half()
    lat = torch.from_numpy(d["latents"])
    xp = lat.clone().to(DEV)
    aux = ops.guidance_update(gn.to(DEV), x.to(DEV), xp, 1, h * h, float(d["beta"]))
    ref = torch.from_numpy(d["out"])
    upd, upd_ref = xp.cpu() - lat, ref - lat
    r, _ = report(f"guidance update vs reference h{h}", upd, upd_ref)
    expect = math.sqrt(2.0) * float((x - lat).norm()) * 1.6
    assert abs(float(upd.norm()) - expect) / expect < 2e-3
    # Measured 2.81e-2 (h = 8) / 2.07e-2 (h = 16); bound = measured x 1.5.  tools/lgp_rounding_floor.py (8 seeds, CPU, the
    # reference imported) separates what this is made of: ANY two fp16 realisations of this ReLU -> BatchNorm x 4 MLP differ
    # in ~0.006 % of their ReLU gates and by 2-4 % in direction (oracle-as-written vs reference 3.1 %, re-associated layer 0
    # vs reference 3.7 %, fp64 truth vs reference 3.7 %), while the moved rounding point of the re-associated layer 0 with
    # the gates held fixed is 0.11 % - the floor is the gates, not the re-association.
    assert r < 4.2e-2
    assert float((upd * upd_ref).sum() / (upd.norm() * upd_ref.norm())) > 0.9991


def test_lgp_forward_backward_vs_oracle_two_samples():
    """S = 2 independent samples (per-sample BatchNorm statistics) at the TINY tap widths, h = 32."""
    from oracle import guidance as og, lgp as olgp, unet as ounet
    from sketch2img_amd.lgp import HipLGP, LOSS_SCALE
    cfg = ounet.TINY
    chans, h, S = ounet.tap_channels(cfg), 32, 2
    sizes = tap_sizes(h)
    sd = olgp.init_state_dict(sum(chans) + 40, seed=11)
    g = torch.Generator().manual_seed(4)
    feats = [[torch.randn(2, c, s, s, generator=g).half().float() for c, s in zip(chans, sizes)] for _ in range(S)]
    noise = torch.randn(S, 4, h, h, generator=g)
    target = 0.2 * torch.randn(S, 4, h, h, generator=g)
    sigma = 0.8
    # HIP: rows = [uncond s0, uncond s1, cond s0, cond s1]
    lgp = HipLGP(sd, chans, DEV)
    taps = []
    for i, s in enumerate(sizes):
        rows = torch.cat([feats[0][i][:1], feats[1][i][:1], feats[0][i][1:], feats[1][i][1:]])
        taps.append((nhwc16(rows), s))
    keep = {}
    out = lgp.forward(taps, noise.to(DEV), sigma, S, h, keep)
    grads, loss = lgp.backward(out, target.to(DEV), keep)
    for smp in range(S):
        fs = [f.clone().requires_grad_(True) for f in feats[smp]]
        x = torch.cat([F.interpolate(f, size=h, mode="bilinear") for f in fs], 1)
        nl = (sigma * noise[smp:smp + 1])
        o = olgp.lgp_forward(sd, x, torch.cat([nl] * 2), training=True)
        oc = o.reshape(2, h, h, 4).permute(0, 3, 2, 1)              # "(b w h) c -> b c h w"
        got = out[:, :4].float().cpu().reshape(2 * S, h, h, 4).permute(0, 3, 1, 2)
        for j in range(2):
            _, m = report(f"lgp out s{smp} row{j}", got[j * S + smp], oc[j])
            assert m <= 6 * 2 ** -8
        ls = F.mse_loss(target[smp:smp + 1], oc[1:2])
        assert abs(float(loss[smp]) - float(ls)) < 2e-3 * max(1.0, float(ls))
        gs = torch.autograd.grad(ls, fs)
        # feature gradients: per-tap relative error.  Bound 8 %: fp16 gate flips of the 4 ReLU layers
        # (see test_guidance_step_matches_reference) + fp16 storage of every backward activation.
        num = den = 0.0
        for i, s in enumerate(sizes):
            gg = from_nhwc(grads[i], S, s, s)[smp] / LOSS_SCALE
            num += float((gg - gs[i][1]).norm() ** 2)
            den += float(gs[i][1].norm() ** 2)
        r = math.sqrt(num / den)
        print(f"[parity] lgp feature-grad s{smp}: rel={r:.3e}")
        assert r < 0.08


# ------------------------------------------------------------------------------------------------------ UNet
@pytest.fixture(scope="module")
def tiny():
    from oracle import unet as ounet
    from sketch2img_amd.config import TINY
    from sketch2img_amd.unet import HipUNet
    cfg = ounet.TINY
    assert vars(cfg) == vars(TINY)
    W = ounet.init_weights(cfg)
    S, h = 2, 32
    g = torch.Generator().manual_seed(21)
    ehs = torch.randn(2 * S, 77, cfg.cross_attention_dim, generator=g).half().float()
    net = HipUNet(TINY, W, DEV)
    net.prepare_context(ehs)
    x = torch.randn(S, 4, h, h, generator=g)
    return dict(cfg=cfg, W=W, net=net, ehs=ehs, x=x, S=S, h=h)


def test_unet_tiny_forward_vs_oracle(tiny):
    from oracle import unet as ounet
    from sketch2img_amd import ops
    from sketch2img_amd.unet import CIN_PAD
    S, h, net = tiny["S"], tiny["h"], tiny["net"]
    xx = torch.cat([tiny["x"], tiny["x"]]).half().float()
    eps, taps = net.forward(ops.nchw_to_nhwc(xx.to(DEV), CIN_PAD), 501, 2 * S, h)
    with torch.no_grad():
        re, rt = ounet.unet_forward(tiny["cfg"], tiny["W"], xx, 501, tiny["ehs"])
    # fp16 storage of ~150 intermediate tensors vs the fp32 oracle.  Measured: eps rel 1.19e-3, taps 0.67 - 1.62e-3 - the
    # same as the oracle's own fp16-storage mode costs (tests/test_gpu_configs.py decomposition); bounds = measured x 1.5
    assert report("unet tiny eps", ops.nhwc_to_nchw(eps, 2 * S, 4, h, h).cpu(), re)[0] < 2e-3
    for i, ((tp, s), r) in enumerate(zip(taps, rt)):
        assert s == r.shape[2]
        assert report(f"unet tiny tap{i}", from_nhwc(tp, 2 * S, s, s), r)[0] < 2.5e-3


def test_unet_tiny_backward_vs_oracle(tiny):
    from oracle import unet as ounet
    from sketch2img_amd import ops
    from sketch2img_amd.unet import CIN_PAD, Stash
    S, h, net, cfg = tiny["S"], tiny["h"], tiny["net"], tiny["cfg"]
    xx = torch.cat([tiny["x"], tiny["x"]]).half().float()
    stash = Stash()
    eps, taps = net.forward(ops.nchw_to_nhwc(xx.to(DEV), CIN_PAD), 501, 2 * S, h, stash)
    g = torch.Generator().manual_seed(33)
    tg = [torch.randn(S, t.shape[1], s, s, generator=g).half().float() for t, s in taps]
    dx = net.backward(stash, [nhwc16(t) for t in tg])
    xr = xx.clone().requires_grad_(True)
    _, rt = ounet.unet_forward(cfg, tiny["W"], xr, 501, tiny["ehs"])
    tot = sum((r[S:] * t).sum() for r, t in zip(rt, tg))
    gr = torch.autograd.grad(tot, xr, retain_graph=True)[0][S:]
    got = ops.nhwc_to_nchw(dx, S, 4, h, h).cpu()
    # every backward activation is stored in fp16 and P / dS are fp16 MFMA operands: measured 2.2e-3, bound = x 1.8
    assert report("unet tiny d/dx", got, gr)[0] < 4e-3
    assert dx[:, 4:].abs().max() == 0
    # each tap on its own (catches a mis-routed skip / tap gradient that a sum could hide)
    for i in (0, 2, 3, 4, 5, 6, 8):
        one = [nhwc16(t if j == i else torch.zeros_like(t)) for j, t in enumerate(tg)]
        dxi = ops.nhwc_to_nchw(net.backward(stash, one), S, 4, h, h).cpu()
        gi = torch.autograd.grad((rt[i][S:] * tg[i]).sum(), xr, retain_graph=True)[0][S:]
        assert report(f"unet tiny d/dx tap{i}", dxi, gi)[0] < 6e-3


def test_sampler_tiny_vs_oracle(tiny):
    """4-step guided trajectory (guided on i = 0, 1, 2), two independent samples, vs oracle.sample_one.
    Per step, teacher-forced from the oracle's state so that errors do not compound: the CFG epsilon,
    the norm of the guidance update (pinned by alpha = sqrt(2)*||dx||/||g||*beta) and its direction."""
    from oracle import guidance as og, lgp as olgp, unet as ounet
    from sketch2img_amd.lgp import HipLGP
    from sketch2img_amd.sampler import DDIMTables, HipSampler
    cfg, S, h = tiny["cfg"], tiny["S"], tiny["h"]
    sd = olgp.init_state_dict(sum(ounet.tap_channels(cfg)) + 40, seed=12)
    g = torch.Generator().manual_seed(44)
    target = 0.18215 * torch.randn(S, 4, h, h, generator=g)
    x0 = tiny["x"]
    T = 4
    traces = []
    for smp in range(S):
        tr = []
        og.sample_one(cfg, tiny["W"], sd, tiny["ehs"][[smp, S + smp]], x0[smp:smp + 1], target[smp:smp + 1], T, trace=tr)
        traces.append(tr)
    sampler = HipSampler(tiny["net"], HipLGP(sd, ounet.tap_channels(cfg), DEV))
    tab = DDIMTables.make(T)
    tiny["net"].prepare_timesteps(tab.timesteps.tolist())
    noise = x0.to(DEV)
    for i in range(T):
        x_i = x0 if i == 0 else torch.cat([traces[s][i - 1]["latents"] for s in range(S)])
        xp, eps, aux = sampler.step(x_i.to(DEV).contiguous(), noise, target.to(DEV), tab, i, 7.5, 1.6, want_eps=True)
        for smp in range(S):
            tr = traces[smp][i]
            assert report(f"sampler step{i} s{smp} eps", eps[smp:smp + 1].cpu(), tr["eps"])[0] < 1.4e-2       # CFG eps: x 9.9
            if tr["aux"] is None:
                assert aux is None
                assert report(f"sampler step{i} s{smp} x_prev", xp[smp:smp + 1].cpu(), tr["latents"])[0] < 1e-2
                continue
            upd_ref = float(tr["aux"]["alpha"]) * tr["aux"]["cond_grad"]
            x_unguided = tr["latents"] - upd_ref
            upd = xp[smp:smp + 1].cpu() - x_unguided
            nr = float(upd.norm() / upd_ref.norm())
            cos = float((upd * upd_ref).sum() / (upd.norm() * upd_ref.norm()))
            print(f"[parity] sampler step{i} s{smp} update: |hip|/|oracle|={nr:.4f} cos={cos:.5f} "
                  f"loss hip={float(aux[smp, 3]):.4e} oracle={float(tr['aux']['loss']):.4e}")
            assert abs(nr - 1) < 2e-2
            # direction: UNet backward (fp16, ~2e-3) + the LGP gradient's ReLU-gate sensitivity (2-3 %,
            # same as reference-vs-oracle in tests/test_oracle.py): cos > 0.998 <=> < 6.3 % relative
            assert cos > 0.998
            assert abs(float(aux[smp, 3]) - float(tr["aux"]["loss"])) < 1e-2 * float(tr["aux"]["loss"])
    # free-running loop: finite, guided on the first three steps only
    out = sampler.sample(x0, target, T)
    assert torch.isfinite(out).all()
    assert [a is not None for a in sampler.last_aux] == [True, True, True, False]
    ref = torch.cat([traces[s][-1]["latents"] for s in range(S)])
    # reported only: three guided updates of norm 2.3x the DDIM step each (T = 4) compound the few-%
    # direction differences chaotically (measured 0.27); the per-step bounds above are the parity claim
    report("sampler free-running final latents", out.cpu(), ref)
    # every kernel on the path is deterministic by construction (no atomics, fixed reduction orders): the same
    # trajectory again, from fresh BatchNorm running statistics, is bit-identical (tools/repeat_check.py does the same
    # on the full-size bench workload) - a difference here would be a race
    outs = []
    for _ in range(2):
        s2 = HipSampler(tiny["net"], HipLGP(sd, ounet.tap_channels(cfg), DEV))
        outs.append(s2.sample(x0, target, T).clone())
    assert torch.equal(outs[0], outs[1])


def test_forked_guidance_branch_is_bit_identical(tiny):
    """HipSampler.fork_guidance: LGP + backward-to-input run on a second HIP stream from the moment the ninth tap exists,
    beside the last up block + conv_out + CFG / DDIM on the launch stream.  Same kernels, same operands, per-stream split-K
    workspaces: four guided steps (the allocator pools get reused across steps) equal the in-line order bit for bit, on TINY
    and on the full SD1.5 architecture at 32 x 32 latents."""
    from oracle import lgp as olgp, unet as ounet
    from sketch2img_amd import synthetic
    from sketch2img_amd.config import SD15, tap_channels
    from sketch2img_amd.lgp import HipLGP
    from sketch2img_amd.sampler import DDIMTables, HipSampler
    from sketch2img_amd.unet import HipUNet
    S, h, net = tiny["S"], tiny["h"], tiny["net"]
    sd = olgp.init_state_dict(sum(ounet.tap_channels(tiny["cfg"])) + 40, seed=12)
    g = torch.Generator().manual_seed(6)
    target = 0.18215 * torch.randn(S, 4, h, h, generator=g)
    big = HipUNet(SD15, synthetic.unet_state_dict(SD15), DEV)
    acc = HipUNet(SD15, synthetic.unet_state_dict(SD15), DEV, residual_fp32=True)      # accuracy mode: the pair forward forks too
    sdb = synthetic.lgp_state_dict(sum(tap_channels(SD15)) + 40)
    # (the last case is the bench shape, 8 samples at 64 x 64: the two streams then both run the persistent weight-stationary
    # GEMM, gemmws.hip, whose counted waits a second stream once broke)
    cases = [(net, sd, ounet.tap_channels(tiny["cfg"]), tiny["x"], target, None),
             (big, sdb, tap_channels(SD15), synthetic.initial_latents(0, 2, 32), synthetic.sketch_targets(0, 2, 32), 2),
             (acc, sdb, tap_channels(SD15), synthetic.initial_latents(0, 2, 32), synthetic.sketch_targets(0, 2, 32), 2),
             (big, sdb, tap_channels(SD15), synthetic.initial_latents(0, 8, 64), synthetic.sketch_targets(0, 8, 64), 8)]
    for unet, lsd, chans, x0, tgt, nctx in cases:
        if nctx is not None:
            unet.prepare_context(synthetic.text_embeddings(nctx))
        tab = DDIMTables.make(8)
        outs = []
        for fork in (False, True):
            smp = HipSampler(unet, HipLGP(lsd, chans, DEV))
            smp.fork_guidance = fork
            x = x0.to(DEV).float()
            trace = []
            for i in range(4):
                x, eps, aux = smp.step(x, x0.to(DEV).float(), tgt.to(DEV).float(), tab, i, 7.5, 1.6, want_eps=True)
                trace += [x.clone(), eps.clone(), aux.clone()]
            torch.cuda.synchronize()
            outs.append(trace)
        assert all(bool(torch.isfinite(a).all()) for a in outs[0])
        assert all(torch.equal(a, b) for a, b in zip(*outs))


def test_shared_cfg_prefix_is_bit_identical(tiny):
    """HipUNet.forward(shared_input=True): the text-independent front of the UNet (conv_in, the first ResnetBlock,
    GroupNorm / proj_in / self-attention / LayerNorm 2 / to_q of the first transformer block) is evaluated once for the two
    identical CFG halves (modules/pipeline.py:85 `torch.cat([latents] * 2)`).  Exact in exact arithmetic.  Where the shared
    layers run the same kernel instantiations at half size (TINY) eps, all nine taps, a guided step's update, its auxiliaries
    and the backward-to-input through the half-size stash equal the doubled evaluation bit for bit; on the full SD1.5
    architecture the half-size launches take other instantiations and the results agree to fp16 rounding noise (below)."""
    from oracle import lgp as olgp, unet as ounet
    from sketch2img_amd import ops, synthetic
    from sketch2img_amd.config import SD15
    from sketch2img_amd.lgp import HipLGP
    from sketch2img_amd.sampler import DDIMTables, HipSampler
    from sketch2img_amd.unet import CIN_PAD, HipUNet
    S, h, net = tiny["S"], tiny["h"], tiny["net"]
    xx = torch.cat([tiny["x"], tiny["x"]])
    x32 = ops.nchw_to_nhwc(xx.to(DEV), CIN_PAD)
    e0, t0 = net.forward(x32, 501, 2 * S, h)
    e1, t1 = net.forward(x32, 501, 2 * S, h, shared_input=True)
    assert torch.equal(e0, e1) and all(torch.equal(a[0], b[0]) for a, b in zip(t0, t1))
    # a guided step: forward with stash, LGP, backward through the half-size stash of the shared front
    sd = olgp.init_state_dict(sum(ounet.tap_channels(tiny["cfg"])) + 40, seed=12)
    g = torch.Generator().manual_seed(5)
    target = 0.18215 * torch.randn(S, 4, h, h, generator=g)
    tab = DDIMTables.make(4)
    outs = []
    for share in (False, True):
        smp = HipSampler(net, HipLGP(sd, ounet.tap_channels(tiny["cfg"]), DEV))
        smp.share_cfg_prefix = share
        xp, eps, aux = smp.step(tiny["x"].to(DEV), tiny["x"].to(DEV), target.to(DEV), tab, 0, 7.5, 1.6, want_eps=True)
        outs.append((xp.clone(), eps.clone(), aux.clone()))
    assert all(torch.equal(a, b) for a, b in zip(*outs))
    # the real architecture at its real top-level resolution: the half-size launches of the shared front take other kernel
    # instantiations than the full-size ones (128 x 160 tiles instead of the 256 x 320 ping-pong tile, split-K at 2 rows), whose
    # GroupNorm partial sums / K slices are summed in another order - the two evaluations then are two fp16 evaluations of
    # the same function, as far apart (measured: eps rel 1.25e-3, max 2.2e-3) as either is from the fp32 oracle, and each
    # deterministic
    W = synthetic.unet_state_dict(SD15)
    big = HipUNet(SD15, W, DEV, need_backward=False)
    for rows in (4, 16):
        big.prepare_context(synthetic.text_embeddings(rows // 2))
        xb = synthetic.initial_latents(0, rows // 2, 64)
        x32 = ops.nchw_to_nhwc(torch.cat([xb, xb]).to(DEV), CIN_PAD)
        e0, t0 = big.forward(x32, 981, rows, 64)
        e1, t1 = big.forward(x32, 981, rows, 64, shared_input=True)
        e2, _ = big.forward(x32, 981, rows, 64, shared_input=True)
        same = torch.equal(e0, e1) and all(torch.equal(a[0], b[0]) for a, b in zip(t0, t1))
        rel = float((e0.float() - e1.float()).norm() / e0.float().norm())
        print(f"[parity] sd15 shared CFG prefix vs doubled evaluation, {rows} rows: bit-identical {same}, eps rel {rel:.3e}, "
              f"max |d eps| {float((e0.float() - e1.float()).abs().max()):.3e}")
        assert rel < 2e-3 and torch.equal(e1, e2)


def test_accuracy_mode_runs_guided_steps():
    """HipUNet(residual_fp32=True) is not forward-only: the pair forward stashes the fp16 (hi) activations and the ordinary
    backward-to-input runs on them.  Full SD1.5 (the pair epilogue needs channel counts that are multiples of 64), one
    sketch, 32x32 latents, the first guided step vs the oracle's trace: the CFG eps is closer to the fp32 oracle than the
    default mode's, the guidance update keeps its norm and direction."""
    from oracle import guidance as og, unet as ounet
    from sketch2img_amd import synthetic
    from sketch2img_amd.config import SD15, tap_channels
    from sketch2img_amd.lgp import HipLGP
    from sketch2img_amd.sampler import DDIMTables, HipSampler
    from sketch2img_amd.unet import HipUNet
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfg = ounet.SD15
    W = synthetic.unet_state_dict(SD15)
    sd = synthetic.lgp_state_dict(synthetic.lgp_input_dim(SD15))
    ehs = synthetic.text_embeddings(1)
    h, T = 32, 2
    x0, tgt = synthetic.initial_latents(0, 1, h), synthetic.sketch_targets(0, 1, h)
    tr = []
    og.sample_one(cfg, W, dict(sd), ehs, x0, tgt, T, trace=tr)
    ref = tr[0]
    tab = DDIMTables.make(T)
    errs = {}
    for mode in (False, True):
        net = HipUNet(SD15, W, DEV, residual_fp32=mode)
        net.prepare_context(ehs)
        sampler = HipSampler(net, HipLGP(sd, tap_channels(SD15), DEV))
        xp, eps, aux = sampler.step(x0.to(DEV), x0.to(DEV), tgt.to(DEV), tab, 0, 7.5, 1.6, want_eps=True)
        errs[mode] = report(f"guided step, residual_fp32={mode}: CFG eps", eps.cpu(), ref["eps"])[0]
        upd_ref = float(ref["aux"]["alpha"]) * ref["aux"]["cond_grad"]
        upd = xp.cpu() - (ref["latents"] - upd_ref)
        nr = float(upd.norm() / upd_ref.norm())
        cos = float((upd * upd_ref).sum() / (upd.norm() * upd_ref.norm()))
        print(f"[parity] guided step, residual_fp32={mode}: |hip|/|oracle|={nr:.4f} cos={cos:.5f}")
        assert abs(nr - 1) < 2e-2 and cos > 0.997
        del net, sampler
    assert errs[True] < 0.8 * errs[False]


def test_unet_sd15_forward_vs_oracle_full_size():
    """One full-size SD1.5 evaluation (2 rows, 64x64 latent, 860 M parameters) vs the fp32 CPU oracle."""
    from oracle import unet as ounet
    from sketch2img_amd import ops
    from sketch2img_amd.config import SD15
    from sketch2img_amd.unet import CIN_PAD, HipUNet
    cfg = ounet.SD15
    W = ounet.init_weights(cfg)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, 4, 64, 64, generator=g).half().float()
    xx = torch.cat([x, x])
    ehs = torch.randn(2, 77, 768, generator=g).half().float()
    net = HipUNet(SD15, W, DEV, need_backward=False)
    net.prepare_context(ehs)
    eps, taps = net.forward(ops.nchw_to_nhwc(xx.to(DEV), CIN_PAD), 981, 2, 64)
    with torch.no_grad():
        re, rt = ounet.unet_forward(cfg, W, xx, 981, ehs)
    # measured: eps rel 1.09e-3 / max 1.94e-3, taps rel 0.65 - 1.58e-3; bounds = measured x 1.5.  The decomposition
    # (HIP vs fp16-storage oracle vs fp32 oracle) is tests/test_gpu_configs.py::test_sd15_eps_error_decomposition_*
    r, m = report("unet sd15 eps", ops.nhwc_to_nchw(eps, 2, 4, 64, 64).cpu(), re)
    assert r < 2e-3 and m < 3e-3
    for i, ((tp, s), r) in enumerate(zip(taps, rt)):
        assert report(f"unet sd15 tap{i}", from_nhwc(tp, 2, s, s), r)[0] < 2.5e-3


def test_sd15_full_size_guided_step_and_backward_vs_oracle():
    """BASELINE's full size (SD1.5, 64x64 latents, 9320-channel LGP), one sample: the UNet backward-to-input
    for random tap gradients, then one complete guided step (i = 0, t = 981), against the CPU oracle."""
    import os
    from oracle import ddim as oddim, guidance as og, unet as ounet
    from sketch2img_amd import ops, synthetic
    from sketch2img_amd.config import SD15, tap_channels
    from sketch2img_amd.lgp import HipLGP
    from sketch2img_amd.sampler import DDIMTables, HipSampler
    from sketch2img_amd.unet import CIN_PAD, HipUNet, Stash
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    cfg = ounet.SD15
    W = synthetic.unet_state_dict(SD15)
    sd = synthetic.lgp_state_dict(synthetic.lgp_input_dim(SD15))
    h = 64
    x0 = synthetic.initial_latents(0, 1, h)
    target = synthetic.sketch_targets(0, 1, h)
    ehs = synthetic.text_embeddings(1)
    net = HipUNet(SD15, W, DEV)
    net.prepare_context(ehs)
    tab = DDIMTables.make(50)
    t = int(tab.timesteps[0])
    net.prepare_timesteps([t])
    # ---- backward-to-input with random tap gradients
    xx = torch.cat([x0, x0]).half().float()
    stash = Stash()
    eps, taps = net.forward(ops.nchw_to_nhwc(xx.to(DEV), CIN_PAD), t, 2, h, stash)
    g = torch.Generator().manual_seed(77)
    tg = [torch.randn(1, tp.shape[1], s, s, generator=g).half().float() for tp, s in taps]
    dx = net.backward(stash, [nhwc16(v) for v in tg])
    xr = xx.clone().requires_grad_(True)
    re, rt = ounet.unet_forward(cfg, W, xr, t, ehs)
    gr = torch.autograd.grad(sum((r[1:] * v).sum() for r, v in zip(rt, tg)), xr, retain_graph=True)[0][1:]
    # measured 2.2e-3 (every backward activation stored in fp16), bound = x 1.8
    assert report("sd15 d/dx (random tap grads)", ops.nhwc_to_nchw(dx, 1, 4, h, h).cpu(), gr)[0] < 4e-3
    del stash, dx
    # ---- one full guided step
    sampler = HipSampler(net, HipLGP(sd, tap_channels(SD15), DEV))
    xp, e_hip, aux = sampler.step(x0.to(DEV), x0.to(DEV), target.to(DEV), tab, 0, 7.5, 1.6, want_eps=True)
    otab = oddim.make_tables(50)
    eu, ec = re.detach().chunk(2)
    e_ref = eu + 7.5 * (ec - eu)
    nxt = oddim.ddim_step(otab, e_ref, t, x0)
    new, oaux = og.apply_anti_gradient(rt, sd, otab.alphas_cumprod, xr, nxt, x0, t, target, 1.6, return_aux=True)
    # CFG-combined eps = eps_u + 7.5 (eps_c - eps_u): the combination amplifies the two rows' independent fp16 errors
    # by ~sqrt(7.5^2 + 6.5^2) = 9.9 (measured 9.2e-3 = 8.4 x the single-row 1.09e-3); bound = measured x 1.5
    assert report("sd15 guided step eps", e_hip.cpu(), e_ref)[0] < 1.4e-2
    upd_ref, upd = new - nxt, xp.cpu() - nxt
    nr = float(upd.norm() / upd_ref.norm())
    cos = float((upd * upd_ref).sum() / (upd.norm() * upd_ref.norm()))
    print(f"[parity] sd15 guided step: |update| hip/oracle={nr:.4f} cos={cos:.5f} loss hip={float(aux[0, 3]):.4e} "
          f"oracle={float(oaux['loss']):.4e}")
    assert abs(nr - 1) < 2e-2 and cos > 0.995
    assert abs(float(aux[0, 3]) - float(oaux["loss"])) < 2e-2 * float(oaux["loss"])


# ------------------------------------------------------------------------------------------------ LGP training step
def test_lgp_training_step_vs_oracle_autograd():
    """One training step (trainer.py:208-252 as intended): loss and the gradient of every LGP parameter vs PyTorch
    autograd through the oracle LGP (fp16-emulating forward, ONE BatchNorm batch of B*h*h rows), then one AdamW
    update vs torch.optim.AdamW on the same gradients, and the running-stat side effects."""
    from oracle import lgp as olgp, unet as ounet
    from sketch2img_amd.lgp import LOSS_SCALE
    from sketch2img_amd.lgp_train import HipLGPTrainer, TRAINABLE
    cfg = ounet.TINY
    chans, h, B = ounet.tap_channels(cfg), 16, 4
    sizes = tap_sizes(h)
    sd = olgp.init_state_dict(sum(chans) + 40, seed=13)
    gen = torch.Generator().manual_seed(6)
    feats = [torch.randn(B, c, s, s, generator=gen).half().float() for c, s in zip(chans, sizes)]
    noise_level = 0.7 * torch.randn(B, 4, h, h, generator=gen)
    target = 0.2 * torch.randn(B, 4, h, h, generator=gen)
    tr = HipLGPTrainer(sd, chans, DEV, lr=2e-4, warmup_steps=0)
    taps = [(nhwc16(f), s) for f, s in zip(feats, sizes)]
    loss, g = tr.loss_and_grads(taps, noise_level, target)
    # ---- oracle: autograd w.r.t. the parameters
    params = {k: sd[k].clone().float().requires_grad_(True) for k in TRAINABLE}
    full = dict(sd)
    full.update(params)
    upd = {k: v.clone() for k, v in sd.items() if "running" in k or "num_batches" in k}
    x = torch.cat([F.interpolate(f, size=h, mode="bilinear") for f in feats], 1)
    o = olgp.lgp_forward(full, x, noise_level, training=True, update_running=upd)
    res = o.reshape(B, h, h, 4).permute(0, 3, 2, 1)                     # "(b w h) c -> b c h w"
    ref_loss = F.mse_loss(res, target)
    ref_g = torch.autograd.grad(ref_loss, [params[k] for k in TRAINABLE])
    assert abs(float(loss) - float(ref_loss)) < 2e-3 * max(1.0, float(ref_loss))
    worst = 0.0
    for k, rg in zip(TRAINABLE, ref_g):
        got = tr.grad_view(g, k).cpu() / LOSS_SCALE
        r = float((got - rg).norm() / (rg.norm() + 1e-12))
        worst = max(worst, r)
        print(f"[parity] lgp dL/d {k:22s} rel={r:.3e} |ref|={float(rg.norm()):.3e}")
        # same bound as the feature gradients: fp16 ReLU-gate flips + fp16 storage of the backward activations
        assert r < 0.08, k
    # running statistics (momentum 0.1, unbiased variance) follow the oracle's
    assert (tr.running_mean[0].cpu() - upd["layers.2.running_mean"]).abs().max() < 2e-3
    assert (tr.running_var[3].cpu() - upd["layers.11.running_var"]).abs().max() < 2e-3
    # ---- AdamW on the HIP gradients vs torch.optim.AdamW on the same numbers
    p0 = {k: sd[k].clone().float().requires_grad_(True) for k in TRAINABLE}
    opt = torch.optim.AdamW([p0[k] for k in TRAINABLE], lr=2e-4, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2)
    for k in TRAINABLE:
        p0[k].grad = tr.grad_view(g, k).cpu().clone() / LOSS_SCALE
    opt.step()
    tr.step(g)
    new = tr.state_dict()
    for k in TRAINABLE:
        assert (new[k].cpu() - p0[k].detach()).abs().max() < 1e-6 + 1e-5 * float(p0[k].detach().abs().max()), k
        o16, _ = tr.layout[k][0], None
        assert torch.equal(tr.w16(k).float().cpu(), new[k].cpu().half().float())      # fp16 working copy refreshed
    assert sorted(new) == sorted(sd) and int(new["layers.2.num_batches_tracked"]) == 1
    # a second step with the updated weights runs and lowers the loss on the same batch
    losses = [float(loss)]
    for _ in range(25):
        l2, g2 = tr.loss_and_grads(taps, noise_level, target)
        tr.step(g2)
        losses.append(float(l2))
    print("[parity] lgp training losses:", " ".join(f"{v:.4f}" for v in losses[::5]))
    assert losses[-1] < losses[0]


def test_lgp_train_step_end_to_end_tiny(tiny):
    """train_step = add_noise + frozen UNet taps (per-sample timesteps) + LGP loss/grads + AdamW, on the TINY UNet;
    the loss of the first step is checked against the oracle UNet + oracle LGP on the same batch."""
    from oracle import ddim as oddim, lgp as olgp, unet as ounet
    from sketch2img_amd.lgp_train import HipLGPTrainer, add_noise, train_step
    cfg, h, B = tiny["cfg"], 32, 2
    chans = ounet.tap_channels(cfg)
    sd = olgp.init_state_dict(sum(chans) + 40, seed=17)
    gen = torch.Generator().manual_seed(9)
    lat = torch.randn(B, 4, h, h, generator=gen)
    sketch = 0.2 * torch.randn(B, 4, h, h, generator=gen)
    noise = torch.randn(B, 4, h, h, generator=gen)
    ts = [801, 133]
    ehs = tiny["ehs"][:B]
    acp = oddim.make_tables(50).alphas_cumprod
    tr = HipLGPTrainer(sd, chans, DEV, warmup_steps=0)
    loss = train_step(tr, tiny["net"], lat, sketch, ehs, ts, noise, acp)
    tiny["net"].prepare_context(tiny["ehs"])                    # restore the fixture's context for later tests
    noisy, nl = add_noise(lat, noise, ts, acp)
    feats = []
    for b in range(B):
        _, taps = ounet.unet_forward(cfg, tiny["W"], noisy[b:b + 1], ts[b], ehs[b:b + 1])
        feats.append(taps)
    x = torch.cat([torch.cat([F.interpolate(f, size=h, mode="bilinear") for f in feats[b]], 1) for b in range(B)])
    o = olgp.lgp_forward(sd, x, nl, training=True)
    ref = F.mse_loss(o.reshape(B, h, h, 4).permute(0, 3, 2, 1), sketch)
    print(f"[parity] lgp train_step loss hip {float(loss):.5f} oracle {float(ref):.5f}")
    assert abs(float(loss) - float(ref)) < 1e-2 * float(ref)
    assert tr.step_count == 1 and int(tr.state_dict()["layers.5.num_batches_tracked"]) == 1


# ------------------------------------------------------------------- declined fused launches fall back to the same numbers
@pytest.mark.parametrize("residual_fp32", [False, True])
def test_declined_conv_shortcut_fold_falls_back_to_two_launches(monkeypatch, residual_fp32):
    """skg_conv3x3_sc_f16 DECLINES (SKG_E_UNSUPPORTED, tests/test_host.py checks the return code of the C ABI on a
    >= 2 GiB operand) and HipUNet._res_fwd / _res_fwd_hp then run conv2 and the shortcut GEMM as two launches from packs that are
    built on first use.  Forced here by making every conv3x3_sc call decline: the evaluation must equal the fused one up to the fp16
    rounding of the shortcut output that the fold removes (default mode) / to pair accuracy (accuracy mode), taps included; the
    accuracy mode's upsampler fall-back keeps the hi-only operand of the launch it replaces."""
    from oracle import unet as ounet
    from sketch2img_amd import ops
    from sketch2img_amd._lib import SkgError
    from sketch2img_amd.config import TINY
    from sketch2img_amd.unet import CIN_PAD, HipUNet
    import dataclasses
    boc = (64, 128, 128, 128)                                   # (the pair kernels want channel counts % 64 == 0)
    cfg, ocfg = dataclasses.replace(TINY, block_out_channels=boc), dataclasses.replace(ounet.TINY, block_out_channels=boc)
    W = ounet.init_weights(ocfg)
    g = torch.Generator().manual_seed(3)
    h, rows = 32, 4
    x = torch.randn(rows, 4, h, h, generator=g)
    ehs = torch.randn(rows, 77, cfg.cross_attention_dim, generator=g).half().float()
    net = HipUNet(cfg, W, DEV, need_backward=False, residual_fp32=residual_fp32)
    assert not residual_fp32 or net.hp_norm_pairs              # (the norm-pair sites of the accuracy mode go through the fall-back too)
    net.prepare_context(ehs)
    lazy_before = set(net.W.lazy)
    assert any(k.endswith(".conv2.weight") for k in lazy_before) and any(k.endswith(".conv_shortcut.weight") for k in lazy_before)

    def run():
        eps, taps = net.forward(ops.nchw_to_nhwc(x.to(DEV), CIN_PAD), 500, rows, h)
        return ops.nhwc_to_nchw(eps, rows, 4, h, h).cpu(), [t[0].float().cpu() for t in taps]

    e0, t0 = run()
    assert set(net.W.lazy) == lazy_before                      # the fused path never touches the fall-back packs
    calls = []

    def declined(*a, **k):
        calls.append(1)
        err = SkgError("forced decline")
        err.rc = -2
        raise err

    monkeypatch.setattr(ops, "conv3x3_sc", declined)
    if residual_fp32:
        monkeypatch.setattr(ops, "conv_up2_pairout", declined)
    e1, t1 = run()
    # (the fall-back built its packs on first use; the norm-pair sites of the accuracy mode slice conv2's taps out of their own folded pack)
    assert len(calls) >= 4 and len(net.W.lazy) < len(lazy_before)
    r, _ = report(f"eps, declined conv3x3_sc vs fused (residual_fp32={residual_fp32})", e1, e0)
    # (accuracy mode: only the blocks of the plain zone - the two deepest levels - round the shortcut output on the fall-back route)
    assert r < (6e-4 if residual_fp32 else 1.5e-3)
    for i, (a, b) in enumerate(zip(t1, t0)):
        assert report(f"tap{i}, declined vs fused", a, b)[0] < 2e-3      # (fp16 taps; the deep ones come from the plain zone in either mode)
    with torch.no_grad():
        C, _ = ounet.unet_forward(ocfg, W, x, 500, ehs)
    assert report("eps, declined path vs fp32 oracle", e1, C)[0] < (8e-4 if residual_fp32 else 2.5e-3)

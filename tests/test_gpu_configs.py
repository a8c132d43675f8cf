"""GPU parity tests (-m gpu) at the BASELINE configurations' REAL sizes:

  * the epsilon-error decomposition of the full SD1.5 evaluation: HIP vs the fp16-storage oracle (kernel error
    proper) and the fp16-storage oracle vs the fp32 oracle (what every fp16 implementation pays);
  * config 4 - SD1.5 @ 64x64 latents with sketch_guided_attn injection (modules/sketch_guided_attn.py:29-40,120-132);
  * config 5 - SD2.1 (heads 5/10/20, head_dim 64, 1024-wide context) @ 96x96 latents with clip_guided_attn on
    [zeros; h] (modules/clip_guided_attn.py:111-125, modules/clip_guided_inf.py:107): 9216 + 257 tokens;
  * a 50-step config-2 trajectory: unguided end latents bounded against the oracle, guided reported + bounded;
  * hipGraph replay == eager, bit for bit;
  * the N > 1 path of bench.py executed as two ranks on one device == the same samples run by one rank, bit for bit.
Everything goes through the C ABI (libskg.so)."""
import json
import math
import os
import subprocess
import sys

import pytest
import torch

from tests.util import report

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _threads():
    torch.set_num_threads(min(32, os.cpu_count() or 1))


def _hip_eps(net, x, t, rows, h):
    from sketch2img_amd import ops
    from sketch2img_amd.unet import CIN_PAD
    eps, taps = net.forward(ops.nchw_to_nhwc(x.to(DEV), CIN_PAD), t, rows, h)
    return ops.nhwc_to_nchw(eps, rows, 4, h, h).cpu(), taps


def _cfg(e):
    eu, ec = e.chunk(2)
    return eu + 7.5 * (ec - eu)


# --------------------------------------------------------------------------------------- epsilon error decomposition
def test_sd15_eps_error_decomposition_hip_vs_fp16_storage_vs_fp32():
    """north_star: <= 1e-3 max latent-eps deviation vs the reference.  The reference's GPU path is fp16
    (app.py:34), so the deviation that can be asked of ANY implementation is the one against an fp16-faithful
    evaluation.  Three evaluations of the same full-size SD1.5 step on identical inputs:
        A  HIP                       (fp16 storage, fp32 accumulation in MFMA, kernel-specific forms)
        B  oracle, fp16 storage      (rounds every stored tensor to fp16, arithmetic in fp32)
        C  oracle, fp32              (no rounding)
    |B - C| is the price of fp16 storage; |A - B| is kernel error + different roundings of nearly equal values;
    |A - C| is what round 1 reported."""
    from oracle import unet as ounet
    from sketch2img_amd.config import SD15
    from sketch2img_amd.unet import HipUNet
    _threads()
    cfg = ounet.SD15
    W = ounet.init_weights(cfg)
    g = torch.Generator().manual_seed(7)
    x = torch.randn(1, 4, 64, 64, generator=g).half().float()
    xx = torch.cat([x, x])
    ehs = torch.randn(2, 77, 768, generator=g).half().float()
    net = HipUNet(SD15, W, DEV, need_backward=False)
    net.prepare_context(ehs)
    A, taps = _hip_eps(net, xx, 981, 2, 64)
    with torch.no_grad():
        C, tC = ounet.unet_forward(cfg, W, xx, 981, ehs)
        with ounet.fp16_storage():
            B, tB = ounet.unet_forward(cfg, W, xx, 981, ehs)
    rAB, mAB = report("sd15 eps  HIP vs fp16-storage oracle", A, B)
    rBC, mBC = report("sd15 eps  fp16-storage oracle vs fp32 oracle", B, C)
    rAC, mAC = report("sd15 eps  HIP vs fp32 oracle", A, C)
    report("sd15 CFG eps (g = 7.5)  HIP vs fp16-storage", _cfg(A), _cfg(B))
    report("sd15 CFG eps (g = 7.5)  fp16-storage vs fp32", _cfg(B), _cfg(C))
    report("sd15 CFG eps (g = 7.5)  HIP vs fp32", _cfg(A), _cfg(C))
    for i, ((tp, s), b, c) in enumerate(zip(taps, tB, tC)):
        a = tp.float().cpu().reshape(2, s, s, -1).permute(0, 3, 1, 2)
        ra, _ = report(f"sd15 tap{i}  HIP vs fp32", a, c)
        rb, _ = report(f"sd15 tap{i}  fp16-storage vs fp32", b, c)
        report(f"sd15 tap{i}  HIP vs fp16-storage", a, b)
        # the HIP path is no further from the fp32 truth than an fp32-arithmetic evaluation with fp16 storage is
        assert ra < 1.25 * rb + 1e-4
    # measured (round 2): |A-C| rel 1.09e-3 / max 1.94e-3; |B-C| rel 1.10e-3 / max 1.50e-3.  Bounds = measured x 1.5.
    assert rAC < 2e-3 and mAC < 3e-3
    assert rAC < 1.25 * rBC and mAC < 1.6 * mBC
    assert rAB < 2e-3 and mAB < 3e-3


def test_sd15_accuracy_mode_meets_north_star_eps_bound():
    """north_star: <= 1e-3 max latent-eps deviation vs the (fp32 CPU) reference.  With every stored tensor in fp16 - the
    reference's own GPU configuration - no implementation gets there (previous test: 1.5-1.8e-3, 1.0e-3 of it from the
    fp16 residual stream alone, tools/eps_decompose.py).  HipUNet(residual_fp32=True) keeps the residual stream and the conv
    outputs that feed a norm / the residual sum as (hi, lo) fp16 pairs (~22 mantissa bits; the pair is the K-doubled operand
    where the stream itself enters a matmul) and - round 6 - the GroupNorm outputs of the last up block whose rounding carries
    the most of what is left (unet.HP_NORM_PAIRS, tools/eps_decompose_sites.py).
    Round 6: 32 rows AT configs[1]'s REAL BATCH - four timesteps across the 50-step schedule, each ONE
    evaluation of all 8 samples (16 rows, the sampler's call form: the instantiations, split factors and the Winograd path of
    the small maps that the timed region of bench.py runs; a 2-row evaluation takes other kernels), samples {0, 2, 5, 7} of each
    against the oracle's evaluation of that sample alone.  The maximum of |eps - eps_fp32| over a row is an extreme value of
    16 384 errors (~4.2 sigma; the worst of 32 rows ~5 sigma, with a tail heavier than Gaussian), so the test prints the
    DISTRIBUTION - per-row maxima, pooled percentiles, sigma - and asserts the north_star number on every row, a margin on the
    pooled percentiles and the scale-free relative distance.  One evaluation is also checked against the oracle's emulation of
    round 5's form of the mode, and one is repeated with conv_out scaled by 2 (exact in fp16): the error scales with the
    output, so the figure for a UNIT-VARIANCE eps is max / std - printed, with the relative bound asserted (DESIGN.md 5
    states the absolute one honestly)."""
    from oracle import unet as ounet
    from sketch2img_amd import ops, synthetic
    from sketch2img_amd.config import SD15
    from sketch2img_amd.unet import CIN_PAD, HipUNet
    _threads()
    cfg = ounet.SD15
    W = synthetic.unet_state_dict(SD15)
    S, h = 8, 64
    lat = synthetic.initial_latents(0, S, h)
    ehs1 = synthetic.text_embeddings(1)
    net = HipUNet(SD15, W, DEV, need_backward=False, residual_fp32=True)
    net.prepare_context(synthetic.text_embeddings(S))
    x16 = ops.nchw_to_nhwc(torch.cat([lat, lat]).to(DEV), CIN_PAD)

    def hip16(t):
        e, _ = net.forward(x16, t, 2 * S, h, want_taps=False, shared_input=True)
        return ops.nhwc_to_nchw(e, 2 * S, 4, h, h).cpu()

    row_max, row_rel, pooled, stds = [], [], [], []
    first = None
    for t in (981, 661, 341, 21):
        g16 = hip16(t)
        for si in (0, 2, 5, 7):
            xx = torch.cat([lat[si:si + 1]] * 2)
            with torch.no_grad():
                C, _ = ounet.unet_forward(cfg, W, xx, t, ehs1)
            A = torch.stack([g16[si], g16[S + si]])
            for row in range(2):
                rAC, mAC = report(f"sd15 eps  HIP accuracy mode (16-row evaluation) vs fp32 oracle, t = {t}, sample {si}, {'cond' if row else 'uncond'} row", A[row], C[row])
                row_max.append(mAC); row_rel.append(rAC); stds.append(float(C[row].std()))
                pooled.append((A[row] - C[row]).abs().flatten())
            if first is None:
                first = (t, si, A, C)
                with torch.no_grad(), ounet.fp16_storage(skip=("res", "lin_n", "rop")):
                    B, _ = ounet.unet_forward(cfg, W, xx, t, ehs1)
                rAC, mAC = report("sd15 eps  HIP accuracy mode vs fp32 oracle", A, C)
                rBC, mBC = report("sd15 eps  oracle emulation of round 5's form of the accuracy mode vs fp32 oracle", B, C)
                report("sd15 eps  HIP accuracy mode vs that emulation", A, B)
                assert mAC < 1.35 * mBC and rAC < 1.25 * rBC + 5e-5
    err = torch.cat(pooled)
    q = lambda f: float(torch.quantile(err[:: max(1, err.numel() // 400000)], f))      # (torch.quantile's input limit)
    sigma = float(err.pow(2).mean().sqrt())
    rm = sorted(row_max)
    worst, worst_rel = rm[-1], max(row_rel)
    print(f"[parity] accuracy mode at the real batch, 4 timesteps x 4 samples x 2 = {len(rm)} rows: per-row max |eps - eps_fp32|: min {rm[0]:.2e} median {rm[len(rm) // 2]:.2e} "
          f"second worst {rm[-2]:.2e} WORST {worst:.2e} (north_star bound 1e-3: margin {100 * (1 - worst / 1e-3):.0f} %); pooled |err|: rms {sigma:.2e} "
          f"p99 {q(0.99):.2e} p99.9 {q(0.999):.2e} max / rms {worst / sigma:.2f}; rel: worst {worst_rel:.2e} mean {sum(row_rel) / len(row_rel):.2e}; "
          f"eps std {min(stds):.3f}-{max(stds):.3f}")
    # measured (round 6, deterministic - fixed reduction orders): worst 7.52e-4, median 6.6e-4, p99 3.8e-4, rel 4.29e-4 mean / 4.43e-4 worst.
    # The bound itself on every row, VERDICT r5's margin line (worst <= 8.5e-4) and the scale-free figure with ~10 % of slack
    assert worst <= 1e-3 and worst <= 8.5e-4 and worst_rel <= 5.0e-4
    assert q(0.99) <= 4.5e-4 and rm[len(rm) // 2] <= 7.5e-4
    # unit-variance report: conv_out x 2 (a power of two: exact) -> every eps and every error doubles; absolute errors for a
    # unit-variance eps are therefore max / std of this model's
    t, si, A, C = first
    saved = {k: net.W[k].clone() for k in ("conv_out.weight:2", "conv_out.bias")}
    for k in saved:
        net.W[k].mul_(2.0)
    g2 = hip16(t)
    for k, v in saved.items():
        net.W[k].copy_(v)
    A2 = torch.stack([g2[si], g2[S + si]])
    lin = float((A2 - 2.0 * A).abs().max())
    r2, m2 = report("sd15 eps  accuracy mode, conv_out x 2 vs 2 x fp32 oracle (linearity of the error in the output scale)", A2, 2.0 * C)
    std = float(C.std())
    print(f"[parity] accuracy mode: conv_out x 2 reproduces 2 x eps to {lin:.1e}; at eps std {2 * std:.2f} the max error reads {m2:.2e}; for a UNIT-VARIANCE "
          f"eps: worst max {worst / min(stds):.2e}, rms {sigma / (sum(stds) / len(stds)):.2e} (absolute), relative {worst_rel:.2e} (scale-free)")
    assert lin <= 4e-6 and abs(m2 / (2.0 * float((A - C).abs().max())) - 1) < 1e-3 and r2 <= 6.2e-4


def test_sd15_accuracy_mode_heavy_tailed_weights():
    """VERDICT r4 next #2, margin evidence: the same bound on weights that are NOT uniform - in every convolution / linear layer 1 %
    of the output channels (at least one) are scaled by 8, the outlier-channel structure trained diffusion UNets show - and four
    more latent seeds (3, 5, 13, 17; with the previous test's 7, 11, 23, 101: eight), at the first and the last timestep of the
    schedule.  The outliers inflate the un-normalised output (|eps| up to 8.7 instead of 1.4: measured, round 5, where the plain
    bound read 4.96e-3 at rel 5.3e-4), so conv_out - the last, linear layer - is rescaled to the OUTPUT SCALE OF THE UNIFORM MODEL
    (max |eps| = 1.4, std 0.37: the scale every number of DESIGN.md section 5 is quoted at); then both the absolute north_star bound
    and the scale-free relative bound are asserted.  (For a unit-variance eps - a trained checkpoint - absolute errors scale by
    1 / 0.37: DESIGN.md section 5 says so.)"""
    from oracle import unet as ounet
    from sketch2img_amd.config import SD15
    from sketch2img_amd.unet import HipUNet
    _threads()
    cfg = ounet.SD15
    W = dict(ounet.init_weights(cfg))
    gsel = torch.Generator().manual_seed(77)
    n_scaled = 0
    for k in sorted(W):
        if k.endswith(".weight") and W[k].dim() >= 2 and "norm" not in k:
            co = W[k].shape[0]
            idx = torch.randperm(co, generator=gsel)[:max(1, co // 100)]
            W[k] = W[k].clone()
            W[k][idx] = (W[k][idx] * 8.0).half().float()
            n_scaled += len(idx)
    cases = [(t, seeds) for t in (981, 21) for seeds in ((3, 5), (13, 17))]

    def inputs(t, seeds):
        g = torch.Generator().manual_seed(seeds[0] * 1000 + t)
        xx = torch.cat([torch.randn(1, 4, 64, 64, generator=torch.Generator().manual_seed(sd_)) for sd_ in seeds]).half().float()
        return xx, torch.randn(2, 77, 768, generator=g).half().float()

    xx, ehs = inputs(*cases[0])
    with torch.no_grad():
        C0, _ = ounet.unet_forward(cfg, W, xx, cases[0][0], ehs)
    amax0 = float(C0.abs().max())
    sc = 2.0 ** round(math.log2(1.4 / amax0))          # a power of two: the rescaled fp16 weights are exact
    W["conv_out.weight"], W["conv_out.bias"] = W["conv_out.weight"] * sc, W["conv_out.bias"] * sc
    net = HipUNet(SD15, W, DEV, need_backward=False, residual_fp32=True)
    worst = worst_rel = amax = 0.0
    for i, (t, seeds) in enumerate(cases):
        xx, ehs = inputs(t, seeds)
        net.prepare_context(ehs)
        A, _ = _hip_eps(net, xx, t, 2, 64)
        if i == 0:
            C = C0 * sc                                  # (conv_out is the last, linear layer: exact)
        else:
            with torch.no_grad():
                C, _ = ounet.unet_forward(cfg, W, xx, t, ehs)
        assert torch.isfinite(A).all()
        amax = max(amax, float(C.abs().max()))
        for row in range(2):
            rAC, mAC = report(f"sd15 eps, heavy-tailed weights: HIP accuracy mode vs fp32 oracle, t = {t}, seed {seeds[row]}", A[row], C[row])
            worst, worst_rel = max(worst, mAC), max(worst_rel, rAC)
    print(f"[parity] accuracy mode, heavy-tailed weights ({n_scaled} output channels x 8; conv_out x {sc:g}: un-normalised max |eps| {amax0:.2f}), "
          f"2 timesteps x 4 seeds: worst max |eps - eps_fp32| = {worst:.2e} (north_star bound 1e-3), worst rel {worst_rel:.2e}, max |eps| {amax:.2f}")
    assert worst <= 1e-3 and worst_rel <= 7e-4 and amax < 2.0


def test_config4_sd15_full_size_sketch_guided_attn_vs_oracle():
    """BASELINE configs[3] at its real size: one CFG-doubled evaluation of the full SD1.5 UNet at 64x64 latents with
    the injected cross-attention on routed residual samples (K / V length N = 4096 / 1024 / 256 / 64)."""
    from oracle import attn_inject, unet as ounet
    from sketch2img_amd import synthetic
    from sketch2img_amd.config import SD15
    from sketch2img_amd.inject import HipInjector
    from sketch2img_amd.unet import HipUNet
    _threads()
    cfg = ounet.SD15
    W = synthetic.unet_state_dict(SD15)
    sd = synthetic.satmixin_state_dict(SD15, "sketch")
    assert all(torch.equal(v, attn_inject.init_state_dict(cfg, "sketch")[k]) for k, v in sd.items())
    h = 64
    x = synthetic.initial_latents(0, 1, h).half().float()
    xx = torch.cat([x, x])
    ehs = synthetic.text_embeddings(1)
    res = synthetic.res_samples(SD15, 0, 1, h)
    net = HipUNet(SD15, W, DEV, need_backward=False)
    net.prepare_context(ehs)
    inj = HipInjector(SD15, sd, "sketch", DEV)
    inj.set_res_samples(res)
    net.inject = inj
    t = 981
    got, _ = _hip_eps(net, xx, t, 2, h)
    with torch.no_grad():
        ref, _ = ounet.unet_forward(cfg, W, xx, t, ehs, inject=attn_inject.make_sketch_inject(cfg, sd, res, 1.0))
        with ounet.fp16_storage():
            ref16, _ = ounet.unet_forward(cfg, W, xx, t, ehs, inject=attn_inject.make_sketch_inject(cfg, sd, res, 1.0))
        base, _ = ounet.unet_forward(cfg, W, xx, t, ehs)
    r, m = report("config 4 eps  HIP vs fp32 oracle", got, ref)
    # the accuracy mode carries the pair stream through the injected attention: north_star's bound on this config too
    acc = HipUNet(SD15, W, DEV, need_backward=False, residual_fp32=True)
    acc.prepare_context(ehs)
    acc.inject = inj
    ga, _ = _hip_eps(acc, xx, t, 2, h)
    ra, ma = report("config 4 eps  HIP accuracy mode vs fp32 oracle", ga, ref)
    assert ma <= 1e-3 and ra <= 7e-4
    del acc
    r16, m16 = report("config 4 eps  HIP vs fp16-storage oracle", got, ref16)
    rs, ms = report("config 4 eps  fp16-storage vs fp32 oracle", ref16, ref)
    assert r < 2.5e-3 and m < 4e-3 and r < 1.3 * rs + 1e-4
    assert float((ref - base).abs().max()) > 1e-2, "the injection must change the output"
    # scale = 0 switches the injection off exactly (h + 0 * conv(a)): equals the plain UNet on the same kernels
    inj.set_scale(0.0)
    off, _ = _hip_eps(net, xx, t, 2, h)
    net.inject = None
    plain, _ = _hip_eps(net, xx, t, 2, h)
    assert report("config 4 scale 0 vs no injection", off, plain)[0] < 1e-3


# ------------------------------------------------------------------------------------------------------ config 5
def test_config5_sd21_768_clip_guided_attn_vs_oracle():
    """BASELINE configs[4] at its real size: SD2.1 architecture (heads 5 / 10 / 20 / 20 = head_dim 64, 1024-wide text
    context, linear projections) at 96x96 latents, second self-attention over 9216 + 257 tokens on [zeros; h]."""
    from oracle import attn_inject, unet as ounet
    from sketch2img_amd import synthetic
    from sketch2img_amd.config import SD21
    from sketch2img_amd.inject import HipInjector
    from sketch2img_amd.unet import HipUNet
    _threads()
    cfg = ounet.SD21
    assert ounet.param_count(cfg) == 865_910_724
    W = synthetic.unet_state_dict(SD21)
    sd = synthetic.satmixin_state_dict(SD21, "clip")
    h = 96
    x = synthetic.initial_latents(0, 1, h).half().float()
    xx = torch.cat([x, x])
    ehs = synthetic.text_embeddings(1, dim=1024)
    state = synthetic.sketch_state(0, 1)                      # [zeros; h]  (clip_guided_inf.py:107)
    assert state.shape == (2, 257, 1024) and float(state[0].abs().max()) == 0.0
    net = HipUNet(SD21, W, DEV, need_backward=False)
    net.prepare_context(ehs)
    inj = HipInjector(SD21, sd, "clip", DEV)
    inj.set_state(state)
    net.inject = inj
    t = 981
    got, _ = _hip_eps(net, xx, t, 2, h)
    with torch.no_grad():
        ref, _ = ounet.unet_forward(cfg, W, xx, t, ehs, inject=attn_inject.make_clip_inject(sd, state, 1.0))
        with ounet.fp16_storage():
            ref16, _ = ounet.unet_forward(cfg, W, xx, t, ehs, inject=attn_inject.make_clip_inject(sd, state, 1.0))
    r, m = report("config 5 eps  HIP vs fp32 oracle", got, ref)
    acc = HipUNet(SD21, W, DEV, need_backward=False, residual_fp32=True)      # accuracy mode: the pair stream through clip_guided_attn
    acc.prepare_context(ehs)
    acc.inject = inj
    ga, _ = _hip_eps(acc, xx, t, 2, h)
    ra, ma = report("config 5 eps  HIP accuracy mode vs fp32 oracle", ga, ref)
    assert ma <= 1e-3 and ra <= 7e-4
    del acc
    report("config 5 eps  HIP vs fp16-storage oracle", got, ref16)
    rs, ms = report("config 5 eps  fp16-storage vs fp32 oracle", ref16, ref)
    assert r < 2.5e-3 and m < 4e-3 and r < 1.3 * rs + 1e-4
    net.inject = None
    plain, _ = _hip_eps(net, xx, t, 2, h)
    assert float((got - plain).abs().max()) > 1e-2, "the injection must change the output"


# ---------------------------------------------------------------------------------- config-2 length trajectories
def test_tiny_50_step_trajectories_unguided_bounded_guided_reported():
    """Config-2 LENGTH (50 DDIM steps, guided on 0..25) on the TINY architecture, free running (no teacher forcing)
    against oracle.sample_one.  Unguided: the end latents stay within a tight bound of the oracle's (errors of the
    fp16 evaluation do not amplify through 50 DDIM steps).  Guided: every update has norm sqrt(2)*||dx||*beta = 2.3x
    the DDIM step and a direction pinned only to cos > 0.998 per step (ReLU-gate floor), so trajectories
    separate; the end-latent distance is reported and bounded loosely, the update norms are bounded tightly."""
    from oracle import guidance as og, lgp as olgp, unet as ounet
    from sketch2img_amd.config import TINY
    from sketch2img_amd.lgp import HipLGP
    from sketch2img_amd.sampler import DDIMTables, HipSampler
    from sketch2img_amd.unet import HipUNet
    _threads()
    cfg = ounet.TINY
    W = ounet.init_weights(cfg)
    sd = olgp.init_state_dict(sum(ounet.tap_channels(cfg)) + 40, seed=12)
    g = torch.Generator().manual_seed(91)
    h, T = 32, 50
    x0 = torch.randn(1, 4, h, h, generator=g)
    ehs = torch.randn(2, 77, cfg.cross_attention_dim, generator=g).half().float()
    target = 0.18215 * torch.randn(1, 4, h, h, generator=g)
    net = HipUNet(TINY, W, DEV)
    net.prepare_context(ehs)
    tab = DDIMTables.make(T)
    # ---- unguided
    sampler = HipSampler(net, None)
    out = sampler.sample(x0, None, T, tables=tab).cpu()
    ref = og.sample_one(cfg, W, sd, ehs, x0, None, T)
    r, m = report("tiny 50-step unguided end latents", out, ref)
    assert r < 5e-3, "unguided 50-step end latents"
    # ---- guided (26 guided steps)
    tr = []
    refg = og.sample_one(cfg, W, dict(sd), ehs, x0, target, T, trace=tr)
    sampler = HipSampler(net, HipLGP(sd, ounet.tap_channels(cfg), DEV))
    outg = sampler.sample(x0, target, T, tables=tab).cpu()
    assert [a is not None for a in sampler.last_aux] == [i <= 25 for i in range(T)]
    rg, _ = report("tiny 50-step guided end latents (reported; chaotic separation, see docstring)", outg, refg)
    assert torch.isfinite(outg).all() and rg < 1.0
    # the guided trajectories stay statistically alike: same overall scale, same final loss level
    assert abs(float(outg.norm() / refg.norm()) - 1) < 0.1
    l_hip = float(sampler.last_aux[25][0, 3])
    l_ref = float(tr[25]["aux"]["loss"])
    print(f"[parity] tiny 50-step guided: loss at the last guided step hip {l_hip:.4e} oracle {l_ref:.4e}")
    assert abs(l_hip - l_ref) < 0.25 * l_ref


# ------------------------------------------------------------------ configs[3] / configs[4] at their REAL batch vs the oracle
@pytest.mark.parametrize("config", [4, 5])
def test_configs_4_and_5_real_batch_vs_oracle(config):
    """The injected-attention configs at the batch bench.py times them at - config 4: SD1.5, 8 samples (16 rows) at 64 x 64 with
    sketch_guided_attn; config 5: SD2.1, 4 samples (8 rows) at 96 x 96 with clip_guided_attn on [zeros; h] - one evaluation in the
    sampler's call form (shared CFG prefix, batched injection K / V, the Winograd path of the 16 x 16 / 8 x 8 resp. 24 x 24 / 12 x 12
    levels), the LAST sample's two rows against the oracle's evaluation of that sample alone; accuracy mode (what `value` is timed in)
    and all-fp16 mode against the same reference."""
    from oracle import attn_inject, unet as ounet
    from sketch2img_amd import ops, synthetic
    from sketch2img_amd.config import SD15, SD21
    from sketch2img_amd.inject import HipInjector
    from sketch2img_amd.unet import CIN_PAD, HipUNet
    _threads()
    c, oc, S, h, dim, variant = (SD15, ounet.SD15, 8, 64, 768, "sketch") if config == 4 else (SD21, ounet.SD21, 4, 96, 1024, "clip")
    W = synthetic.unet_state_dict(c)
    sd = synthetic.satmixin_state_dict(c, variant)
    lat = synthetic.initial_latents(0, S, h)
    si, t = S - 1, 981
    ehs1 = synthetic.text_embeddings(1, dim=dim)
    xx = torch.cat([lat[si:si + 1]] * 2)
    with torch.no_grad():
        if config == 4:
            inj1 = attn_inject.make_sketch_inject(oc, sd, synthetic.res_samples(c, si, 1, h), 1.0)
        else:
            inj1 = attn_inject.make_clip_inject(sd, synthetic.sketch_state(si, 1), 1.0)
        ref, _ = ounet.unet_forward(oc, W, xx, t, ehs1, inject=inj1)
    x32 = ops.nchw_to_nhwc(torch.cat([lat, lat]).to(DEV), CIN_PAD)
    for residual_fp32 in (True, False):
        net = HipUNet(c, W, DEV, need_backward=False, residual_fp32=residual_fp32)
        net.prepare_context(synthetic.text_embeddings(S, dim=dim))
        inj = HipInjector(c, sd, variant, DEV)
        inj.set_scale(1.0)
        if config == 4:
            inj.set_res_samples(synthetic.res_samples(c, 0, S, h))
        else:
            inj.set_state(synthetic.sketch_state(0, S))
        net.inject = inj
        e, _ = net.forward(x32, t, 2 * S, h, want_taps=False, shared_input=True)
        g = ops.nhwc_to_nchw(e, 2 * S, 4, h, h).cpu()
        got = torch.stack([g[si], g[S + si]])
        r, m = report(f"config {config}, {2 * S} rows, residual_fp32={residual_fp32}: eps of sample {si} vs fp32 oracle", got, ref)
        assert torch.isfinite(g).all()
        if residual_fp32:
            assert m <= 1e-3 and r <= 7e-4
        else:
            assert r < 2.5e-3 and m < 4e-3
        del net, inj
        torch.cuda.empty_cache()


# ------------------------------------------------------------------ configs[1] at its REAL batch: 16 rows vs the oracle
@pytest.mark.parametrize("residual_fp32", [False, True])
def test_sd15_config1_real_batch_16_rows_vs_oracle(residual_fp32):
    """every other full-size oracle comparison runs 2 rows (M = 8 192 at the 64 x 64 level); the bench runs
    configs[1]'s real batch - 8 samples = 16 rows, M = 65 536 - where dispatch picks other instantiations (the 256 x 320 ping-pong
    tile, no split-K at 64 x 64 / 32 x 32, other split factors below).  Here: full SD1.5 @ 64 x 64, 8 samples, (1) one evaluation
    at t = 981 and (2) one guided step (i = 0 of 50) through the sampler, both modes; rows {0, 8} and {7, 15} - samples 0 and 7,
    the first and the last of the batch - against oracle.unet.unet_forward / oracle.guidance.apply_anti_gradient on those
    two samples ALONE (samples are independent: the reference itself only runs B = 1, modules/pipeline.py:160).  Same bounds
    as the 2-row tests.  Prints the instantiations the 16-row launches walk beside the 2-row ones."""
    from oracle import ddim as oddim, guidance as og, unet as ounet
    from sketch2img_amd import ops, synthetic
    from sketch2img_amd._lib import lib
    from sketch2img_amd.config import SD15, tap_channels
    from sketch2img_amd.lgp import HipLGP
    from sketch2img_amd.sampler import DDIMTables, HipSampler
    from sketch2img_amd.unet import CIN_PAD, HipUNet
    _threads()
    cfg = ounet.SD15
    S, h, T = 8, 64, 50
    W = synthetic.unet_state_dict(SD15)
    sd = synthetic.lgp_state_dict(synthetic.lgp_input_dim(SD15))
    x0, tgt = synthetic.initial_latents(0, S, h), synthetic.sketch_targets(0, S, h)
    net = HipUNet(SD15, W, DEV, residual_fp32=residual_fp32)
    net.prepare_context(synthetic.text_embeddings(S))
    tab, otab = DDIMTables.make(T), oddim.make_tables(T)
    t = int(tab.timesteps[0])
    net.prepare_timesteps(tab.timesteps.tolist())
    shapes = [("conv 64x64 320->320", 4096, 320, 2880, 320, 1), ("conv 32x32 640->640", 1024, 640, 5760, 640, 1),
              ("conv 16x16 1280->1280", 256, 1280, 11520, 1280, 1), ("conv 8x8 1280->1280", 64, 1280, 11520, 1280, 1),
              ("gemm 64x64 K320 N320", 4096, 320, 320, 0, 0), ("gemm 64x64 QKV", 4096, 960, 320, 0, 0),
              ("gemm 16x16 K1280 N1280", 256, 1280, 1280, 0, 0), ("gemm 16x16 FF2 K5120", 256, 1280, 5120, 0, 0)]
    for name, hw, N, K, Cin, mode in shapes:
        print(f"[parity] instantiation (skg_gemm_variant) {name:24s}: 16 rows -> {lib.skg_gemm_variant(16 * hw, N, K, Cin, mode)}, "
              f"8 rows (cond-only backward) -> {lib.skg_gemm_variant(8 * hw, N, K, Cin, mode)}, 2 rows -> {lib.skg_gemm_variant(2 * hw, N, K, Cin, mode)}")
    assert lib.skg_gemm_variant(16 * 4096, 320, 2880, 320, 1) != lib.skg_gemm_variant(2 * 4096, 320, 2880, 320, 1)
    # (1) one evaluation, 16 rows, the sampler's own call form (shared CFG prefix)
    x32 = ops.nchw_to_nhwc(torch.cat([x0, x0]).to(DEV), CIN_PAD)
    eps, taps = net.forward(x32, t, 2 * S, h, shared_input=True)
    A = ops.nhwc_to_nchw(eps, 2 * S, 4, h, h).cpu()
    taps_hip = [(tp.float().cpu().reshape(2 * S, s_, s_, -1).permute(0, 3, 1, 2), s_) for tp, s_ in taps]
    # (2) one guided step of the 8 samples
    sampler = HipSampler(net, HipLGP(sd, tap_channels(SD15), DEV))
    xp, eps_cfg, aux = sampler.step(x0.to(DEV), x0.to(DEV), tgt.to(DEV), tab, 0, 7.5, 1.6, want_eps=True)
    xp, eps_cfg, aux = xp.cpu(), eps_cfg.cpu(), aux.cpu()
    assert torch.isfinite(xp).all()
    ehs1 = synthetic.text_embeddings(1)
    for si in (0, S - 1):
        xi, ti = x0[si:si + 1], tgt[si:si + 1]
        x_in = torch.cat([xi] * 2).requires_grad_(True)
        with torch.enable_grad():
            eo, to = ounet.unet_forward(cfg, W, x_in, t, ehs1)
        eu, ec = eo.detach().chunk(2)
        e_cfg = eu + 7.5 * (ec - eu)
        nxt = oddim.ddim_step(otab, e_cfg, t, xi)
        new, ao = og.apply_anti_gradient(to, dict(sd), otab.alphas_cumprod, x_in, nxt, xi, t, ti, 1.6, return_aux=True)
        for row, orow in ((si, 0), (S + si, 1)):
            r, m = report(f"sd15 16 rows, residual_fp32={residual_fp32}: eps row {row} (sample {si}) vs fp32 oracle", A[row], eo.detach()[orow])
            if residual_fp32:
                assert m <= 1e-3 and r <= 7e-4              # north_star's bound, as the 2-row sweep asserts it
            else:
                assert r < 2e-3 and m < 3e-3                # the default mode's 2-row bounds
            for k, ((th, s_), tor) in enumerate(zip(taps_hip, to)):
                rt = float((th[row] - tor.detach()[orow]).norm() / tor.detach()[orow].norm())
                assert rt < 2.5e-3, (k, row, rt)      # (fp16 taps; the deep ones come from plain-fp16 blocks in either mode)
        rc, _ = report(f"sd15 16 rows, residual_fp32={residual_fp32}: CFG eps of the guided step, sample {si}", eps_cfg[si], e_cfg[0])
        assert rc < (7e-3 if residual_fp32 else 1.4e-2)
        upd_ref = float(ao["alpha"]) * ao["cond_grad"]
        upd = xp[si:si + 1] - (new - upd_ref)
        nr = float(upd.norm() / upd_ref.norm())
        cos = float((upd * upd_ref).sum() / (upd.norm() * upd_ref.norm()))
        dl = abs(float(aux[si, 3]) - float(ao["loss"])) / float(ao["loss"])
        print(f"[parity] sd15 16 rows, residual_fp32={residual_fp32}: guided step, sample {si}: |hip|/|oracle| = {nr:.5f}, cos = {cos:.5f}, "
              f"loss hip {float(aux[si, 3]):.4e} oracle {float(ao['loss']):.4e}")
        # free-running within the step (the HIP update starts from the HIP x_{t-1}): the scheduler step's own eps error enters `upd`
        assert abs(nr - 1) < 2e-2 and cos > 0.9965 and dl < 2e-3


def test_accuracy_mode_pair_zone_winograd_option(monkeypatch):
    """unet._HP_WINO (SKG_HP_WINO=1, off by default): Winograd also for the ResnetBlock convolutions of the accuracy mode's PAIR zone at the
    16 x 16 level - pair output through the output transform, the K-doubled shortcut as a pair GEMM whose result is the residual.  One 16-row
    evaluation at the real batch: close to the shipped form (two fp16 realisations of the 16 x 16 convolutions) and, sample 0 against the fp32
    oracle, inside the mode's bound; measured trade (profiles/r06_eps_real_batch_hp_wino.txt): rel 4.27e-4 -> 4.56e-4, worst row of 48
    7.9e-4 -> 8.8e-4, - 2.2 % per evaluation, + 1.5 % end to end - on the accuracy / time line, past the margin line, hence an option."""
    from oracle import unet as ounet
    from sketch2img_amd import ops, synthetic, unet as hunet
    from sketch2img_amd.config import SD15
    from sketch2img_amd.unet import CIN_PAD, HipUNet
    _threads()
    S, h, t = 8, 64, 981
    W = synthetic.unet_state_dict(SD15)
    lat = synthetic.initial_latents(0, S, h)
    x16 = ops.nchw_to_nhwc(torch.cat([lat, lat]).to(DEV), CIN_PAD)
    outs = []
    for on in (False, True):
        monkeypatch.setattr(hunet, "_HP_WINO", on)
        net = HipUNet(SD15, W, DEV, need_backward=False, residual_fp32=True)
        assert any(k.startswith("down_blocks.2.") and k.endswith(":wino") for k in net.W) == on
        net.prepare_context(synthetic.text_embeddings(S))
        e, _ = net.forward(x16, t, 2 * S, h, want_taps=False, shared_input=True)
        outs.append(ops.nhwc_to_nchw(e, 2 * S, 4, h, h).cpu())
        del net
        torch.cuda.empty_cache()
    r, _ = report("sd15 accuracy mode, 16 rows: Winograd in the pair zone vs the shipped form", outs[1], outs[0])
    assert torch.isfinite(outs[1]).all() and 1e-5 < r < 6e-4
    with torch.no_grad():
        C, _ = ounet.unet_forward(ounet.SD15, W, torch.cat([lat[:1]] * 2), t, synthetic.text_embeddings(1))
    A = torch.stack([outs[1][0], outs[1][S]])
    ra, ma = report("sd15 accuracy mode, Winograd in the pair zone, sample 0 vs fp32 oracle", A, C)
    assert ra < 5.5e-4 and ma < 1e-3


# ------------------------------------------------------------------ full architecture, config[0]'s trajectory shape
def test_sd15_config0_trajectories_vs_oracle():
    """VERDICT r2 missing #4: trajectory-level parity on the REAL architecture.  BASELINE configs[0]'s shape - full SD1.5
    (860 M parameters), one sketch, 256x256 (32x32 latents), 10 DDIM steps, guidance on steps 0..5 - against
    oracle.guidance.sample_one (the loop of modules/pipeline.py:83-115):
      (a) UNGUIDED, free running: end latents of the HIP loop vs the oracle's, tight bound (per-step errors do not amplify);
      (b) GUIDED, teacher-forced per step from the oracle's trace (so that the chaotic separation of guided trajectories,
          DESIGN.md 5, does not enter): CFG eps, the norm of the update (alpha = sqrt(2) ||dx|| / ||g|| * beta), its
          direction, the loss - at every one of the 10 steps, 6 of them guided."""
    from oracle import guidance as og, unet as ounet
    from sketch2img_amd import synthetic
    from sketch2img_amd.config import SD15, tap_channels
    from sketch2img_amd.lgp import HipLGP
    from sketch2img_amd.sampler import DDIMTables, HipSampler
    from sketch2img_amd.unet import HipUNet
    _threads()
    cfg = ounet.SD15
    W = synthetic.unet_state_dict(SD15)
    sd = synthetic.lgp_state_dict(synthetic.lgp_input_dim(SD15))
    ehs = synthetic.text_embeddings(1)
    h, T = 32, 10
    x0, tgt = synthetic.initial_latents(0, 1, h), synthetic.sketch_targets(0, 1, h)
    net = HipUNet(SD15, W, DEV)
    net.prepare_context(ehs)
    tab = DDIMTables.make(T)
    # (a) unguided, free running
    with torch.no_grad():
        ref_u = og.sample_one(cfg, W, None, ehs, x0, None, T)
    out_u = HipSampler(net, None).sample(x0, None, T, tables=tab).cpu()
    ru, mu = report("sd15 config[0] unguided 10-step end latents, free running", out_u, ref_u)
    assert torch.isfinite(out_u).all() and ru < 2.2e-3          # measured 1.41e-3 (round 3), x 1.5
    # (b) guided, teacher-forced - two samples (initial latents 1000 + i, sketch 2000 + i)
    sampler = HipSampler(net, HipLGP(sd, tap_channels(SD15), DEV))
    net.prepare_timesteps(tab.timesteps.tolist())
    worst = dict(eps=0.0, nr=0.0, cos=1.0, loss=0.0)
    omc = []                                             # 1 - cos of every guided step
    for si in (0, 1):
        x0, tgt = synthetic.initial_latents(si, 1, h), synthetic.sketch_targets(si, 1, h)
        tr = []
        og.sample_one(cfg, W, dict(sd), ehs, x0, tgt, T, trace=tr)
        assert [t["aux"] is not None for t in tr] == [i <= 5 for i in range(T)]              # Q6: i <= 0.5 T
        noise = x0.to(DEV)
        for i in range(T):
            x_i = x0 if i == 0 else tr[i - 1]["latents"]
            xp, eps, aux = sampler.step(x_i.to(DEV).contiguous(), noise, tgt.to(DEV), tab, i, 7.5, 1.6, want_eps=True)
            e, _ = report(f"sd15 config[0] sample {si} step{i} CFG eps (teacher-forced)", eps.cpu(), tr[i]["eps"])
            worst["eps"] = max(worst["eps"], e)
            assert e < 1.4e-2                                # CFG-combined eps: 8.4 x the single-row error
            if tr[i]["aux"] is None:
                assert aux is None
                assert report(f"sd15 config[0] sample {si} step{i} x_prev", xp.cpu(), tr[i]["latents"])[0] < 2e-3
                continue
            upd_ref = float(tr[i]["aux"]["alpha"]) * tr[i]["aux"]["cond_grad"]
            upd = xp.cpu() - (tr[i]["latents"] - upd_ref)
            nr = float(upd.norm() / upd_ref.norm())
            cos = float((upd * upd_ref).sum() / (upd.norm() * upd_ref.norm()))
            dl = abs(float(aux[0, 3]) - float(tr[i]["aux"]["loss"])) / float(tr[i]["aux"]["loss"])
            print(f"[parity] sd15 config[0] sample {si} step{i} update: |hip|/|oracle|={nr:.4f} cos={cos:.5f} "
                  f"loss hip={float(aux[0, 3]):.4e} oracle={float(tr[i]['aux']['loss']):.4e}")
            worst["nr"], worst["cos"], worst["loss"] = max(worst["nr"], abs(nr - 1)), min(worst["cos"], cos), max(worst["loss"], dl)
            omc.append(1.0 - cos)
            # |ratio - 1| <= 1.5e-5 and loss rel <= 4.1e-4 measured (round 3): bounds = measured x 1.5 and more.  The DIRECTION: round 4
            # fitted cos > 0.9965 to one step of one sample (0.99765 with the stashing cross-attention launch, 0.99916 without).  Round 5
            # measured both paths over 8 samples x 6 guided steps (tools/traj_seeds.py, profiles/r05_traj_direction_seeds.txt):
            # 1 - cos mean 5.98e-4 / max 1.64e-3 with the fused launch, mean 6.75e-4 / max 2.73e-3 with the per-operator launches - the
            # same distribution (the unfused path holds the worst step of the 96): single steps scatter up to ~3e-3 in either
            # realisation, which is the LGP's rounding floor (profiles/r03_lgp_rounding_floor.txt), not a property of the launch.
            # So: every step above the floor 0.9965, and the MEAN over the guided steps at round 3's level (1 - cos < 1.1e-3).
            assert abs(nr - 1) < 1e-3 and cos > 0.9965 and dl < 2e-3
    mean_omc = sum(omc) / len(omc)
    print(f"[parity] sd15 config[0] guided, worst over 2 samples x 10 steps: eps rel {worst['eps']:.2e}, | |upd| ratio - 1 | {worst['nr']:.2e}, "
          f"cos {worst['cos']:.5f} (mean 1 - cos {mean_omc:.2e} over {len(omc)} guided steps), loss rel {worst['loss']:.2e}")
    assert mean_omc < 1.1e-3
    # (c) round 6: the same two checks in the ACCURACY mode - the mode bench.py's `value` is timed in.  Unguided free running: the end latents
    # follow the oracle's 3-4 x closer than the all-fp16 mode's; guided, teacher-forced, the last sample's trace: CFG eps, update norm / direction / loss
    del net, sampler
    torch.cuda.empty_cache()
    acc = HipUNet(SD15, W, DEV, residual_fp32=True)
    acc.prepare_context(ehs)
    out_a = HipSampler(acc, None).sample(synthetic.initial_latents(0, 1, h), None, T, tables=tab).cpu()
    ra, _ = report("sd15 config[0] unguided 10-step end latents, free running, ACCURACY mode", out_a, ref_u)
    assert torch.isfinite(out_a).all() and ra < 8e-4 and ra < 0.6 * ru
    smp = HipSampler(acc, HipLGP(sd, tap_channels(SD15), DEV))
    acc.prepare_timesteps(tab.timesteps.tolist())
    noise = x0.to(DEV)
    w_eps, w_cos, w_nr = 0.0, 1.0, 0.0
    for i in range(T):
        x_i = x0 if i == 0 else tr[i - 1]["latents"]
        xp, eps, aux = smp.step(x_i.to(DEV).contiguous(), noise, tgt.to(DEV), tab, i, 7.5, 1.6, want_eps=True)
        e, _ = report(f"sd15 config[0] ACCURACY mode sample 1 step{i} CFG eps (teacher-forced)", eps.cpu(), tr[i]["eps"])
        w_eps = max(w_eps, e)
        if tr[i]["aux"] is None:
            assert report(f"sd15 config[0] ACCURACY mode step{i} x_prev", xp.cpu(), tr[i]["latents"])[0] < 8e-4
            continue
        upd_ref = float(tr[i]["aux"]["alpha"]) * tr[i]["aux"]["cond_grad"]
        upd = xp.cpu() - (tr[i]["latents"] - upd_ref)
        w_nr = max(w_nr, abs(float(upd.norm() / upd_ref.norm()) - 1))
        w_cos = min(w_cos, float((upd * upd_ref).sum() / (upd.norm() * upd_ref.norm())))
    print(f"[parity] sd15 config[0] ACCURACY mode: unguided end latents rel {ra:.2e} (all-fp16: {ru:.2e}); guided, teacher-forced, 10 steps: worst CFG eps rel {w_eps:.2e}, "
          f"| |upd| ratio - 1 | {w_nr:.2e}, cos {w_cos:.5f}")
    assert w_eps < 7e-3 and w_nr < 1e-3 and w_cos > 0.9965


@pytest.mark.parametrize("case", ["dpm", "sketch", "clip"])
def test_full_architecture_free_running_trajectories_other_configs(case):
    """Free-running multi-step parity on the REAL architectures for the paths the previous test does not walk:
      dpm    - SD1.5, DPM-Solver++ 2M (the scheduler app.py:13-25 actually builds), 6 unguided steps, 32x32 latents;
      sketch - SD1.5 with sketch_guided_attn injection (BASELINE configs[3]), 4 DDIM steps, 32x32 latents;
      clip   - the SD2.1 architecture with clip_guided_attn on [zeros; h] (configs[4]), 3 DDIM steps, 32x32 latents
               (1024 + 257 tokens in the injected attention).
    End latents of the HIP loop vs oracle.guidance.sample_one on identical inputs (loop: modules/pipeline.py:83-115;
    injection: modules/sketch_guided_attn.py:120-132, modules/clip_guided_attn.py:111-125)."""
    from oracle import attn_inject, guidance as og, unet as ounet
    from sketch2img_amd import synthetic
    from sketch2img_amd.config import SD15, SD21
    from sketch2img_amd.inject import HipInjector
    from sketch2img_amd.sampler import DDIMTables, DPMTables, HipSampler
    from sketch2img_amd.unet import HipUNet
    _threads()
    h = 32
    hcfg, ocfg = (SD21, ounet.SD21) if case == "clip" else (SD15, ounet.SD15)
    W = synthetic.unet_state_dict(hcfg)
    ehs = synthetic.text_embeddings(1, dim=hcfg.cross_attention_dim)
    x0 = synthetic.initial_latents(0, 1, h)
    net = HipUNet(hcfg, W, DEV, need_backward=False)
    net.prepare_context(ehs)
    inject, T, tab, sched = None, 4, None, "ddim"
    if case == "dpm":
        T, sched = 6, "dpm++2m"
        tab = DPMTables.make(T)
    elif case == "sketch":
        sd = synthetic.satmixin_state_dict(SD15, "sketch")
        res = synthetic.res_samples(SD15, 0, 1, h)
        inj = HipInjector(SD15, sd, "sketch", DEV)
        inj.set_res_samples(res)
        net.inject = inj
        inject = attn_inject.make_sketch_inject(ocfg, sd, res, 1.0)
    else:
        T = 3
        sd = synthetic.satmixin_state_dict(SD21, "clip")
        state = synthetic.sketch_state(0, 1)
        inj = HipInjector(SD21, sd, "clip", DEV)
        inj.set_state(state)
        net.inject = inj
        inject = attn_inject.make_clip_inject(sd, state, 1.0)
    tab = tab or DDIMTables.make(T)
    with torch.no_grad():
        ref = og.sample_one(ocfg, W, None, ehs, x0, None, T, inject=inject, scheduler=sched)
    out = HipSampler(net, None).sample(x0, None, T, tables=tab).cpu()
    r, _ = report(f"{case}: {T}-step free-running end latents on the full architecture", out, ref)
    assert torch.isfinite(out).all() and r < 3e-3


# ----------------------------------------------------------------------------------------------- hipGraph replay
@pytest.mark.parametrize("sched", ["ddim", "dpm"])
def test_graph_replay_is_bit_identical_to_eager(sched):
    from oracle import lgp as olgp, unet as ounet
    from sketch2img_amd.config import TINY
    from sketch2img_amd.lgp import HipLGP
    from sketch2img_amd.sampler import DDIMTables, DPMTables, HipSampler
    from sketch2img_amd.unet import HipUNet
    cfg = ounet.TINY
    W = ounet.init_weights(cfg)
    sd = olgp.init_state_dict(sum(ounet.tap_channels(cfg)) + 40, seed=12)
    g = torch.Generator().manual_seed(3)
    h, T, S = 32, 6, 2
    x0 = torch.randn(S, 4, h, h, generator=g)
    ehs = torch.randn(2 * S, 77, cfg.cross_attention_dim, generator=g).half().float()
    target = 0.18215 * torch.randn(S, 4, h, h, generator=g)
    net = HipUNet(TINY, W, DEV)
    net.prepare_context(ehs)
    tab = (DPMTables if sched == "dpm" else DDIMTables).make(T)
    eager = HipSampler(net, HipLGP(sd, ounet.tap_channels(cfg), DEV))
    a = eager.sample(x0, target, T, tables=tab).clone()
    lg = HipLGP(sd, ounet.tap_channels(cfg), DEV)
    graphed = HipSampler(net, lg, use_graphs=True)
    b = graphed.sample(x0, target, T, tables=tab).clone()        # capture + first replay
    c = graphed.sample(x0, target, T, tables=tab).clone()        # replay only
    assert torch.equal(a, b) and torch.equal(a, c)
    for x, y in zip(eager.last_aux, graphed.last_aux):
        assert (x is None) == (y is None) and (x is None or torch.equal(x, y))
    # BatchNorm side effects match the eager run's after the same number of trajectories (2 here vs 1 eager -> compare counts)
    assert lg.num_batches_tracked == [2 * n for n in eager.lgp.num_batches_tracked]


@pytest.mark.parametrize("residual_fp32", [False, True])
def test_graph_replay_full_size_with_winograd_is_bit_identical(residual_fp32):
    """Round 6: the Winograd path (three launches per convolution, 16 fp32 slabs in the stream's workspace, the GroupNorm that writes the
    input transform) and the accuracy mode's mixed zones under hipGraph capture, at configs[1]'s real size - full SD1.5, 8 samples, 64 x 64
    latents, 3 DDIM steps of which 2 are guided (the cond-only backward through the Winograd data gradients): replay == eager, bit for bit,
    and a second replay reproduces it."""
    from sketch2img_amd import synthetic
    from sketch2img_amd.config import SD15, tap_channels
    from sketch2img_amd.lgp import HipLGP
    from sketch2img_amd.sampler import DDIMTables, HipSampler
    from sketch2img_amd.unet import HipUNet
    S, h, T = 8, 64, 3
    W = synthetic.unet_state_dict(SD15)
    sd = synthetic.lgp_state_dict(synthetic.lgp_input_dim(SD15))
    x0, tgt = synthetic.initial_latents(0, S, h), synthetic.sketch_targets(0, S, h)
    net = HipUNet(SD15, W, DEV, residual_fp32=residual_fp32)
    assert any(k.endswith(":wino") for k in net.W) and any(k.endswith(":winoT") for k in net.W)
    net.prepare_context(synthetic.text_embeddings(S))
    tab = DDIMTables.make(T)
    eager = HipSampler(net, HipLGP(sd, tap_channels(SD15), DEV))
    a = eager.sample(x0, tgt, T, tables=tab).clone()
    graphed = HipSampler(net, HipLGP(sd, tap_channels(SD15), DEV), use_graphs=True)
    b = graphed.sample(x0, tgt, T, tables=tab).clone()        # capture + first replay
    c = graphed.sample(x0, tgt, T, tables=tab).clone()        # replay only
    assert torch.isfinite(a).all() and torch.equal(a, b) and torch.equal(a, c)
    assert [x is not None for x in eager.last_aux] == [True, True, False]
    for x, y in zip(eager.last_aux, graphed.last_aux):
        assert (x is None) == (y is None) and (x is None or torch.equal(x, y))


@pytest.mark.parametrize("variant", ["clip", "sketch"])
def test_graph_replay_with_injected_attention_is_bit_identical(variant):
    """bench.py --graph is allowed with configs 4 / 5: the injectors' per-image buffers are created by the eager
    warm-up pass, the captured steps only read / refresh them."""
    from oracle import unet as ounet
    from sketch2img_amd import synthetic
    from sketch2img_amd.config import TINY
    from sketch2img_amd.inject import HipInjector
    from sketch2img_amd.sampler import DDIMTables, HipSampler
    from sketch2img_amd.unet import HipUNet
    W = ounet.init_weights(ounet.TINY)
    g = torch.Generator().manual_seed(4)
    h, T, S = 32, 4, 2
    x0 = torch.randn(S, 4, h, h, generator=g)
    ehs = torch.randn(2 * S, 77, TINY.cross_attention_dim, generator=g).half().float()
    net = HipUNet(TINY, W, DEV, need_backward=False)
    net.prepare_context(ehs)
    inj = HipInjector(TINY, synthetic.satmixin_state_dict(TINY, variant), variant, DEV)
    if variant == "clip":
        inj.set_state(synthetic.sketch_state(0, S))
    else:
        inj.set_res_samples(synthetic.res_samples(TINY, 0, S, h))
    net.inject = inj
    tab = DDIMTables.make(T)
    a = HipSampler(net, None).sample(x0, None, T, tables=tab).clone()
    gs = HipSampler(net, None, use_graphs=True)
    b = gs.sample(x0, None, T, tables=tab).clone()
    c = gs.sample(x0, None, T, tables=tab).clone()
    assert torch.equal(a, b) and torch.equal(a, c)
    # the shared CFG front (HipUNet.forward(shared_input=True), on by default in the sampler) vs the doubled evaluation:
    # for the sketch variant with equal halves of the res samples it runs THROUGH the injection, for the clip variant
    # ([zeros; h]) it ends in front of it; TINY runs the same kernel instantiations at half size -> bit-identical
    assert inj.halves_equal == (variant == "sketch")
    off = HipSampler(net, None)
    off.share_cfg_prefix = False
    assert torch.equal(off.sample(x0, None, T, tables=tab), a)
    net.inject = None
    plain = HipSampler(net, None).sample(x0, None, T, tables=tab)
    assert not torch.equal(plain, a)


def test_graph_cache_follows_prompt_sketch_and_scale_changes():
    """a captured step points at the prompt's K / V and the injector's per-image K / V.  A second image with
    another prompt, another sketch or another scale must not replay the old graphs: every graphed call equals the eager
    call made in the same state, the cache stays bounded, and going back to an earlier state re-captures (no stale hit)."""
    from oracle import unet as ounet
    from sketch2img_amd import synthetic
    from sketch2img_amd.config import TINY
    from sketch2img_amd.inject import HipInjector
    from sketch2img_amd.sampler import DDIMTables, HipSampler
    from sketch2img_amd.unet import HipUNet
    W = ounet.init_weights(ounet.TINY)
    g = torch.Generator().manual_seed(5)
    h, T, S = 32, 3, 1
    x0 = torch.randn(S, 4, h, h, generator=g)
    prompts = [torch.randn(2 * S, 77, TINY.cross_attention_dim, generator=g).half().float() for _ in range(2)]
    net = HipUNet(TINY, W, DEV, need_backward=False)
    inj = HipInjector(TINY, synthetic.satmixin_state_dict(TINY, "sketch"), "sketch", DEV)
    net.inject = inj
    tab = DDIMTables.make(T)
    gs = HipSampler(net, None, use_graphs=True)
    gs.max_graph_sets = 2
    seen = []
    for prompt, sketch, scale in [(0, 0, 1.0), (1, 0, 1.0), (1, 3, 1.0), (1, 3, 0.5), (0, 0, 1.0)]:
        net.prepare_context(prompts[prompt])
        inj.set_res_samples(synthetic.res_samples(TINY, sketch, S, h))
        inj.set_scale(scale)
        e = HipSampler(net, None).sample(x0, None, T, tables=tab).clone()
        a = gs.sample(x0, None, T, tables=tab).clone()
        b = gs.sample(x0, None, T, tables=tab).clone()          # replay of the graphs captured for THIS state
        assert torch.equal(e, a) and torch.equal(e, b), (prompt, sketch, scale)
        seen.append(e)
        assert len(gs._graphs) <= 2
    assert not torch.equal(seen[0], seen[1]) and not torch.equal(seen[1], seen[2]) and not torch.equal(seen[2], seen[3])
    assert torch.equal(seen[0], seen[4])


# ------------------------------------------------------------------------------------- N > 1 on real hardware
def _bench(args, env=None, nproc=1, port=29611, launcher=False):
    cmd = [sys.executable]
    if nproc > 1 or launcher:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
                "--master-port", str(port)]
    cmd += [os.path.join(ROOT, "bench.py")] + args
    e = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    e.update(env or {})
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=1500, env=e, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_on_one_device_equal_one_rank_bitwise(tmp_path):
    """bench.py --gpus 2 as two processes on cuda:0 (gloo carries the collectives): broadcast_state_dict -> HipUNet ->
    sample -> on-rank VAE decode -> gather of uint8 images.  The gathered images of global samples {0, 1} must equal,
    bit for bit, what a single rank produces for sample 0 and for sample 1 (placement independence)."""
    common = ["--steps", "1", "--warmup", "0", "--ddim-steps", "3", "--samples-per-gpu", "1", "--no-cpu-baseline",
              "--no-roofline", "--no-box-probe", "--no-second-mode"]
    two = tmp_path / "two.pt"
    # (round 6: plain `python bench.py --gpus 2` - bench.py launches its own ranks under torch.distributed.run)
    d2 = _bench(common + ["--gpus", "2", "--dump-images", str(two)],
                env={"SKG_BENCH_BACKEND": "gloo", "SKG_BENCH_DEVICE": "0"})
    assert d2["n_gpus"] == 2 and d2["config"]["global_batch"] == 2 and d2["outputs_finite"]
    assert d2["out_shape"] == [2, 512, 512, 3]
    got = torch.load(two)["images"]
    assert got.dtype == torch.uint8
    for i in range(2):
        one = tmp_path / f"one{i}.pt"
        d1 = _bench(common + ["--gpus", "1", "--first-sample", str(i), "--dump-images", str(one)])
        assert d1["n_gpus"] == 1 and d1["out_shape"] == [1, 512, 512, 3]
        ref = torch.load(one)["images"]
        assert torch.equal(got[i:i + 1], ref), f"global sample {i}: 2-rank gather differs from the 1-rank result"
    assert not torch.equal(got[0], got[1])


def test_rccl_path_executes_on_one_gpu(tmp_path):
    """VERDICT r2 missing #2: the `nccl` backend (= RCCL) had never executed.  bench.py under torch.distributed.run with ONE
    rank and SKG_BENCH_FORCE_DIST=1 initialises the process group on the device and sends the weight broadcast
    (broadcast_state_dict: UNet / LGP / VAE buckets, int64 buffers included) and the final gather of the decoded uint8 images
    through real RCCL calls on device tensors; the gathered images must equal the plain single-process run bit for bit."""
    common = ["--steps", "1", "--warmup", "0", "--ddim-steps", "3", "--samples-per-gpu", "2", "--no-cpu-baseline",
              "--no-roofline", "--gpus", "1", "--no-box-probe", "--no-second-mode"]
    a, b = tmp_path / "rccl.pt", tmp_path / "plain.pt"
    d = _bench(common + ["--dump-images", str(a)], env={"SKG_BENCH_FORCE_DIST": "1"}, launcher=True, port=29633)
    assert "backend nccl = RCCL" in d["config"]["parallelism"] and len(d["config"]["parallelism"]) <= 120 and d["outputs_finite"] and d["out_shape"] == [2, 512, 512, 3]
    p = _bench(common + ["--dump-images", str(b)])
    assert "no process group" in p["config"]["parallelism"]
    ia, ib = torch.load(a), torch.load(b)
    assert ia["images"].dtype == torch.uint8 and torch.equal(ia["images"], ib["images"]) and torch.equal(ia["latents"], ib["latents"])


@pytest.mark.parametrize("config", [4, 5])
def test_bench_contract_line_configs_4_and_5(config):
    d = _bench(["--config", str(config), "--steps", "1", "--warmup", "0", "--ddim-steps", "2", "--samples-per-gpu", "1",
                "--no-cpu-baseline"])
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["value"] > 0 and d["outputs_finite"] and d["config"]["baseline_config"] == config
    assert f"configs[{config - 1}]" in d["config"]["workload"] and "model" not in d["config"]
    assert d["config"]["mode"] == "residual_fp32" and d["config"]["fast_fp16_value"] > 0 and d["config"]["box_mfma_tflops"] > 0
    assert d["out_shape"] == ([1, 512, 512, 3] if config == 4 else [1, 768, 768, 3])
    roof = d["roofline"]
    assert roof["bound"] == "mfma" and 0 < roof["frac"] < 1 and roof["unit"] == "TFLOP/s"
    assert abs(d["tflop_per_image"] * 25 - (98.97 if config == 4 else 300.19)) < 0.02     # 2 of 50 steps

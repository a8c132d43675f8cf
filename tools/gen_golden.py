#!/usr/bin/env python3
"""Generate golden vectors by running the REFERENCE's own Python in the build container.

Runs ONLY where /root/reference exists (never on the GPU box; nothing under tests/ or the
product imports this file).  It injects a names-only stub ``diffusers`` (no arithmetic: SURVEY.md
Appendix A) so that the reference's modules/latent_predictor.py and modules/pipeline.py import,
then calls the reference's LatentEdgePredictor, get_noise_level and apply_anti_gradient on seeded
inputs and stores inputs + outputs under tests/golden/ as small .npz files.

    python tools/gen_golden.py            # rewrites tests/golden/*.npz, *.json
"""
import json
import logging
import os
import sys
import types

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def import_reference():
    d = types.ModuleType("diffusers")
    d.UNet2DConditionModel = object
    d.StableDiffusionPipeline = type("StableDiffusionPipeline", (), {})
    du = types.ModuleType("diffusers.utils")
    du.logging = types.SimpleNamespace(get_logger=logging.getLogger)
    sys.modules["diffusers"], sys.modules["diffusers.utils"] = d, du
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    from modules.latent_predictor import LatentEdgePredictor, hook_unet  # noqa
    from modules.pipeline import AntiGradientPipeline  # noqa
    return LatentEdgePredictor, hook_unet, AntiGradientPipeline


TAP_C = [32] * 9                                  # toy tap widths (multiples of the GEMM K slice)


def tap_sizes(h):
    return [h // 2, h // 4, h // 8, h // 8, h // 8, h // 8, h // 4, h // 2, h]


def sd_to_np(sd):
    return {k: v.detach().cpu().numpy().copy() for k, v in sd.items()}   # copy: .numpy() aliases live buffers


def seeded_lgp(LatentEdgePredictor, input_dim, seed):
    torch.manual_seed(seed)
    m = LatentEdgePredictor(input_dim, 4, 9)
    g = torch.Generator().manual_seed(seed + 1)
    with torch.no_grad():                     # move BN affine off (1,0) so it is exercised
        for mod in m.layers:
            if isinstance(mod, torch.nn.BatchNorm1d):
                mod.weight.add_(0.2 * (torch.rand(mod.weight.shape, generator=g) - 0.5))
                mod.bias.add_(0.2 * (torch.rand(mod.bias.shape, generator=g) - 0.5))
    return m.half()                           # fp16 weights: required by latent_predictor.py:43 (Q4)


def main():
    torch.set_num_threads(1)
    os.makedirs(OUT, exist_ok=True)
    LatentEdgePredictor, hook_unet, AntiGradientPipeline = import_reference()

    # 1. checkpoint manifest of the real-size module ------------------------------------------
    m = LatentEdgePredictor(9320, 4, 9)
    manifest = {k: [list(v.shape), str(v.dtype)] for k, v in m.state_dict().items()}
    meta = dict(manifest=manifest, n_params=sum(p.numel() for p in m.parameters()),
                default_training=bool(m.training))
    del m

    # 2. LGP forward, train-mode and eval-mode BN ----------------------------------------------
    for h in (8, 16):
        C = 128
        lgp = seeded_lgp(LatentEdgePredictor, C + 40, seed=100 + h)
        sd0 = sd_to_np(lgp.state_dict())
        g = torch.Generator().manual_seed(7 + h)
        x = torch.randn(2, C, h, h, generator=g)
        t0 = 0.7 * torch.randn(1, 4, h, h, generator=g)
        t = torch.cat([t0] * 2)                     # as the pipeline passes it: cat([noise_level] * 2)
        lgp.train()
        with torch.no_grad():
            y_train = lgp(x, t)
        sd1 = sd_to_np(lgp.state_dict())            # running stats after ONE train-mode call (Q3)
        lgp.load_state_dict({k: torch.from_numpy(v) for k, v in sd0.items()})
        lgp.eval()
        with torch.no_grad():
            y_eval = lgp(x, t)
        np.savez_compressed(os.path.join(OUT, f"lgp_fwd_h{h}.npz"), x=x.numpy(), t=t.numpy(),
                            y_train=y_train.float().numpy(), y_eval=y_eval.float().numpy(),
                            **{"sd." + k: v for k, v in sd0.items()},
                            **{"sd_after." + k: v for k, v in sd1.items() if "running" in k or "num_batches" in k})

    # 3+4. get_noise_level and apply_anti_gradient on a differentiable toy feature extractor ----
    betas = torch.linspace(0.00085 ** 0.5, 0.012 ** 0.5, 1000, dtype=torch.float32) ** 2
    acp = torch.cumprod(1 - betas, 0)
    for h, tstep in ((8, 981), (16, 501)):
        p = AntiGradientPipeline.__new__(AntiGradientPipeline)
        p.scheduler = types.SimpleNamespace(alphas_cumprod=acp)
        lgp = seeded_lgp(LatentEdgePredictor, sum(TAP_C) + 40, seed=300 + h)
        lgp.train()
        p.lgp_model = lgp
        sd0 = sd_to_np(lgp.state_dict())
        g = torch.Generator().manual_seed(11 + h)
        convs = [0.5 * torch.randn(c, 4, 1, 1, generator=g) for c in TAP_C]
        x = torch.randn(1, 4, h, h, generator=g)
        x_in = torch.cat([x] * 2).requires_grad_(True)
        latents = x + 0.1 * torch.randn(1, 4, h, h, generator=g)     # stands in for x_{t-1}
        noise = torch.randn(1, 4, h, h, generator=g)
        target = 0.18215 * torch.randn(1, 4, h, h, generator=g)
        # noise level incl. the fp16-noise -> fp32 promotion (Q5)
        nl16 = p.get_noise_level(noise.half(), torch.tensor(tstep))
        assert nl16.dtype == torch.float32
        blocks = []
        with torch.enable_grad():
            for w, s in zip(convs, tap_sizes(h)):
                f = torch.tanh(F.conv2d(x_in, w))
                f = F.adaptive_avg_pool2d(f, s)
                blocks.append(types.SimpleNamespace(output=f.float()))
            p.feature_blocks = blocks
            out = p.apply_anti_gradient(x_in, latents, noise, torch.tensor(tstep), target, 1.6)
        np.savez_compressed(
            os.path.join(OUT, f"guidance_h{h}.npz"), x=x.numpy(), latents=latents.numpy(),
            noise=noise.numpy(), target=target.numpy(), t=np.int64(tstep), beta=np.float32(1.6),
            out=out.detach().float().numpy(), noise_level_fp16_noise=nl16.numpy(),
            alphas_cumprod=acp.numpy(),
            **{f"conv{i}": w.numpy() for i, w in enumerate(convs)},
            **{"sd." + k: v for k, v in sd0.items()})

    # 6. B = 2 raises at pipeline.py:160 (Q1) ---------------------------------------------------
    raised = False
    try:
        h = 8
        p = AntiGradientPipeline.__new__(AntiGradientPipeline)
        p.scheduler = types.SimpleNamespace(alphas_cumprod=acp)
        p.lgp_model = seeded_lgp(LatentEdgePredictor, sum(TAP_C) + 40, seed=1)
        x = torch.randn(2, 4, h, h)
        x_in = torch.cat([x] * 2).requires_grad_(True)
        with torch.enable_grad():
            p.feature_blocks = [types.SimpleNamespace(output=F.adaptive_avg_pool2d(
                torch.tanh(F.conv2d(x_in, torch.randn(c, 4, 1, 1))), s).float())
                for c, s in zip(TAP_C, tap_sizes(h))]
            p.apply_anti_gradient(x_in, x.clone(), torch.randn(2, 4, h, h), torch.tensor(500),
                                  torch.randn(1, 4, h, h), 1.6)
    except RuntimeError as e:
        raised = True
        meta["b2_error"] = str(e)[:120]
    meta["b2_raises"] = raised

    # 5. guided-step index sets, evaluated with the expressions of pipeline.py:89-92,108 ---------
    gs = {}
    for T in (10, 50):
        step_stop = 0.5 * T
        gs[str(T)] = [i for i in range(T) if i <= step_stop and not (i > step_stop)]
    meta["guided_steps"] = gs
    with open(os.path.join(OUT, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print("wrote", sorted(os.listdir(OUT)))


if __name__ == "__main__":
    main()

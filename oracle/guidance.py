"""Oracle: the sketch-guidance step and the sampling loop.  TEST INFRASTRUCTURE.

Follows the reference's modules/pipeline.py:
  * get_noise_level        :132-139   sqrt(1 - alphas_cumprod[t]) * noise, fp32
  * apply_anti_gradient    :141-161   bilinear resize of the 9 taps to (h, h) -> concat -> LGP ->
                                      "(b w h) c -> b c h w" -> cond chunk -> MSE(mean) ->
                                      g = -dLoss/dx_in (cond chunk) ->
                                      alpha = ||x_in - x_{t-1}|| / ||g|| * beta  (norm over BOTH CFG
                                      copies: the sqrt(2) of SURVEY Q2) -> x_{t-1} + alpha g
  * __call__ loop          :83-115    guided iff i <= 0.5*T (Q6); CFG :99-101; noise = initial
                                      latents (:75); guidance applied AFTER scheduler.step (Q7)
apply_anti_gradient / get_noise_level are PINNED against golden vectors made by importing that
file (tests/golden/guidance_*.npz).  The loop additionally uses oracle.unet / oracle.ddim
(third-party arithmetic, PARITY UNPINNED - see oracle/__init__.py).

B > 1 is defined as B independent B=1 runs (the reference crashes for B > 1: SURVEY Q1).
"""
from __future__ import annotations

from typing import Callable, Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from . import ddim as _ddim
from . import dpmsolver as _dpm
from . import lgp as _lgp
from . import unet as _unet


def get_noise_level(alphas_cumprod: torch.Tensor, noise: torch.Tensor, t: int) -> torch.Tensor:
    s = (1 - alphas_cumprod[t]) ** 0.5          # fp32 scalar tensor (CPU table)
    return s.reshape(1, 1, 1, 1) * noise        # promotes fp16 noise to fp32 (Q5)


def apply_anti_gradient(taps: Sequence[torch.Tensor], lgp_sd: Dict[str, torch.Tensor],
                        alphas_cumprod: torch.Tensor, latents_prev: torch.Tensor,
                        latents: torch.Tensor, noise: torch.Tensor, t: int,
                        target: Optional[torch.Tensor], beta: float = 1.6, *,
                        training: bool = True, emulate_fp16: bool = True,
                        update_running=None, compute_dtype=torch.float32, return_aux: bool = False):
    """``taps``: the 9 hooked feature maps (fp32, each (2,C_i,s_i,s_i)) carrying an autograd graph
    back to ``latents_prev`` (the CFG-doubled, requires_grad UNet input).  ``latents`` is
    x_{t-1} from the scheduler, (1,4,h,h)."""
    if target is None:
        return latents
    h = latents.shape[2]
    assert latents.shape[2] == latents.shape[3], "square latents only (SURVEY Q8)"
    resized = [F.interpolate(tp.float() if compute_dtype == torch.float32 else tp.to(compute_dtype),
                             size=h, mode="bilinear") for tp in taps]
    feats = torch.cat(resized, dim=1)
    nl = get_noise_level(alphas_cumprod, noise, t).to(feats.dtype)
    out = _lgp.lgp_forward(lgp_sd, feats, torch.cat([nl] * 2), training=training,
                           emulate_fp16=emulate_fp16, update_running=update_running,
                           compute_dtype=compute_dtype)
    b = latents_prev.shape[0]
    out = out.reshape(b, h, h, -1).permute(0, 3, 2, 1)          # "(b w h) c -> b c h w"
    out_c = out.chunk(2)[1]
    loss = F.mse_loss(target.to(out_c.dtype), out_c, reduction="mean")
    grad = torch.autograd.grad(loss, latents_prev)[0]
    cond_grad = (-grad).chunk(2)[1]
    num = torch.linalg.norm(latents_prev.detach() - latents)    # broadcast over both CFG rows (Q2)
    den = torch.linalg.norm(cond_grad)
    alpha = num / den * beta
    new = latents + alpha * cond_grad
    if return_aux:
        return new.detach(), dict(alpha=alpha.detach(), gnorm=den.detach(), loss=loss.detach(),
                                  out_c=out_c.detach(), cond_grad=cond_grad.detach())
    return new.detach()


def guided_steps(T: int) -> List[int]:
    """indices i with i <= 0.5*T (modules/pipeline.py:89-92,108)."""
    return [i for i in range(T) if not (i > 0.5 * T)]


def sample_one(cfg: _unet.UNetConfig, W: Dict[str, torch.Tensor], lgp_sd, ehs: torch.Tensor,
               latents0: torch.Tensor, target: Optional[torch.Tensor], num_inference_steps: int,
               guidance_scale: float = 7.5, beta: float = 1.6, *, emulate_fp16: bool = True,
               inject=None, trace: Optional[list] = None,
               step_hook: Optional[Callable] = None, scheduler: str = "ddim") -> torch.Tensor:
    """One B=1 trajectory, modules/pipeline.py:83-115.  ``ehs`` is (2,77,D) = [uncond; cond];
    ``latents0`` (1,4,h,h).  Returns the final latents (1,4,h,h)."""
    # scheduler: "ddim" (the BASELINE metric) or "dpm++2m" (what app.py:13-25 configures)
    dpm = scheduler == "dpm++2m"
    assert dpm or scheduler == "ddim"
    tab = _dpm.make_tables(num_inference_steps) if dpm else _ddim.make_tables(num_inference_steps)
    dpm_state = _dpm.DPMState()
    latents = latents0.clone()
    noise = latents0.detach().clone()
    T = len(tab.timesteps)
    for i, t in enumerate(tab.timesteps.tolist()):
        guided = (not (i > 0.5 * T)) and target is not None
        x_in = torch.cat([latents] * 2).detach().requires_grad_(guided)
        with torch.enable_grad() if guided else torch.no_grad():
            eps, taps = _unet.unet_forward(cfg, W, x_in, t, ehs, inject=inject)
        eu, ec = eps.detach().chunk(2)
        e = eu + guidance_scale * (ec - eu)
        nxt = _dpm.dpm_step(tab, dpm_state, e, i, latents) if dpm else _ddim.ddim_step(tab, e, t, latents)
        aux = None
        if guided:
            nxt, aux = apply_anti_gradient(taps, lgp_sd, tab.alphas_cumprod, x_in, nxt, noise, t,
                                           target, beta, emulate_fp16=emulate_fp16, return_aux=True)
        if trace is not None:
            trace.append(dict(i=i, t=t, eps=e.clone(), latents=nxt.clone(), aux=aux))
        if step_hook is not None:
            step_hook(i, t, nxt)
        latents = nxt.detach()
    return latents

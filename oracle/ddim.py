"""Oracle: DDIM scheduler (eta = 0) as the reference drives it.  TEST INFRASTRUCTURE.

Reference call sites: modules/pipeline.py:60-61 (set_timesteps / timesteps), :86
(scale_model_input = identity), :104 (step(...).prev_sample), and the beta schedule at
app.py:15-19 / trainer.py:188-194 (scaled_linear, 0.00085 .. 0.012, 1000 train steps).
The arithmetic itself is third-party diffusers DDIMScheduler (absent here): PARITY UNPINNED,
restated from SURVEY.md section 8 (a2) and checked by known-answer tests (timestep tables,
alphas_cumprod end points).
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch


@dataclass
class DDIMTables:
    alphas_cumprod: torch.Tensor      # (1000,) fp32, CPU
    final_alpha_cumprod: float
    timesteps: np.ndarray             # (T,) int64, descending
    ratio: int


def make_tables(num_inference_steps: int, num_train_timesteps: int = 1000,
                beta_start: float = 0.00085, beta_end: float = 0.012,
                steps_offset: int = 1, set_alpha_to_one: bool = False) -> DDIMTables:
    betas = torch.linspace(beta_start ** 0.5, beta_end ** 0.5, num_train_timesteps,
                           dtype=torch.float32) ** 2
    alphas_cumprod = torch.cumprod(1.0 - betas, dim=0)
    final = 1.0 if set_alpha_to_one else float(alphas_cumprod[0])
    ratio = num_train_timesteps // num_inference_steps
    ts = (np.arange(0, num_inference_steps) * ratio).round()[::-1].copy().astype(np.int64)
    ts = ts + steps_offset
    return DDIMTables(alphas_cumprod, final, ts, ratio)


def step_coeffs(tab: DDIMTables, t: int):
    """(sqrt(abar_t), sqrt(1-abar_t), sqrt(abar_prev), sqrt(1-abar_prev)) for eta=0, as python floats
    computed from the fp32 table exactly like the scheduler does (fp32 tensor scalars)."""
    prev = t - tab.ratio
    a_t = tab.alphas_cumprod[t]
    a_p = tab.alphas_cumprod[prev] if prev >= 0 else torch.tensor(tab.final_alpha_cumprod)
    return (float(a_t ** 0.5), float((1 - a_t) ** 0.5), float(a_p ** 0.5), float((1 - a_p) ** 0.5))


def ddim_step(tab: DDIMTables, eps: torch.Tensor, t: int, x: torch.Tensor, v_prediction: bool = False) -> torch.Tensor:
    """x_{t-1} for eta = 0, no clipping.  `eps` is the model output: epsilon, or - v_prediction (the SD2.1-768 checkpoint's
    scheduler config; published parameterisation of Salimans & Ho, "Progressive distillation", as DDIMScheduler.step
    applies it) - v = sqrt(abar) eps - sqrt(1 - abar) x0, i.e. x0 = sqrt(abar) x - sqrt(1 - abar) v and
    eps = sqrt(abar) v + sqrt(1 - abar) x."""
    prev = t - tab.ratio
    a_t = tab.alphas_cumprod[t].to(x.dtype)
    a_p = (tab.alphas_cumprod[prev] if prev >= 0 else torch.tensor(tab.final_alpha_cumprod)).to(x.dtype)
    if v_prediction:
        v = eps
        x0 = a_t ** 0.5 * x - (1 - a_t) ** 0.5 * v
        eps = a_t ** 0.5 * v + (1 - a_t) ** 0.5 * x
        return a_p ** 0.5 * x0 + (1 - a_p) ** 0.5 * eps
    x0 = (x - (1 - a_t) ** 0.5 * eps) / a_t ** 0.5
    direction = (1 - a_p) ** 0.5 * eps
    return a_p ** 0.5 * x0 + direction

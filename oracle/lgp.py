"""Oracle: Latent Edge/Gradient Predictor (LGP) MLP.  TEST INFRASTRUCTURE.

Follows the reference's modules/latent_predictor.py:9-45 line by line:
  * :39-40  pos = cat_l sin(2*pi*t*2^-l), l = 0..num_layers-1, along channels
  * :42     x = cat(x, t, pos) along channels           (9280 + 4 + 36 = 9320)
  * :43     "b c h w -> (b w h) c", hard cast to fp16    
  * :15-29  Linear -> ReLU -> BatchNorm1d, four times, then Linear(64, out)  (ReLU BEFORE BN)
  * BatchNorm honours ``training`` (default True in the app: SURVEY Q3): batch statistics with
    biased variance for normalisation, eps 1e-5; running stats (momentum 0.1, unbiased variance,
    num_batches_tracked += 1) are side effects.
PINNED against golden vectors made by importing that file (tests/golden/lgp_*.npz).

The checkpoint format is the reference module's ``state_dict`` (30 keys, a5 in SURVEY.md):
``layers.{0,3,6,9,12}.{weight,bias}``, ``layers.{2,5,8,11}.{weight,bias,running_mean,
running_var,num_batches_tracked}``.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, Optional

import torch

LIN = (0, 3, 6, 9, 12)
BNS = (2, 5, 8, 11)
HIDDEN = (512, 256, 128, 64)


def state_dict_manifest(input_dim: int = 9320, output_dim: int = 4) -> "OrderedDict[str, tuple]":
    m: "OrderedDict[str, tuple]" = OrderedDict()
    dims = (input_dim,) + HIDDEN + (output_dim,)
    for i in range(5):
        m[f"layers.{LIN[i]}.weight"] = (dims[i + 1], dims[i])
        m[f"layers.{LIN[i]}.bias"] = (dims[i + 1],)
        if i < 4:
            b = BNS[i]
            m[f"layers.{b}.weight"] = (dims[i + 1],)
            m[f"layers.{b}.bias"] = (dims[i + 1],)
            m[f"layers.{b}.running_mean"] = (dims[i + 1],)
            m[f"layers.{b}.running_var"] = (dims[i + 1],)
            m[f"layers.{b}.num_batches_tracked"] = ()
    return m


def init_state_dict(input_dim: int = 9320, output_dim: int = 4, seed: int = 20260929,
                    perturb_bn: bool = True) -> Dict[str, torch.Tensor]:
    """kaiming-uniform weights / zero bias as modules/latent_predictor.py:32-35 (values rounded
    through fp16).  ``perturb_bn`` moves BN gamma/beta off (1, 0) so they are exercised."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for k, shp in state_dict_manifest(input_dim, output_dim).items():
        idx = int(k.split(".")[1])
        leaf = k.split(".")[2]
        if idx in LIN:
            if leaf == "weight":
                bound = math.sqrt(6.0 / shp[1])          # kaiming_uniform_, a=0, fan_in
                v = (torch.rand(shp, generator=g) * 2 - 1) * bound
            else:
                v = torch.zeros(shp)
        else:
            if leaf == "weight":
                v = torch.ones(shp) + (0.2 * (torch.rand(shp, generator=g) - 0.5) if perturb_bn else 0)
            elif leaf == "bias":
                v = 0.2 * (torch.rand(shp, generator=g) - 0.5) if perturb_bn else torch.zeros(shp)
            elif leaf == "running_mean":
                v = torch.zeros(shp)
            elif leaf == "running_var":
                v = torch.ones(shp)
            else:
                v = torch.zeros((), dtype=torch.int64)
        sd[k] = v.half().float() if v.dtype.is_floating_point else v
    return sd


def positional_channels(t: torch.Tensor, num_layers: int = 9) -> torch.Tensor:
    """modules/latent_predictor.py:39-40.  fp32 in, fp32 out."""
    return torch.cat([torch.sin(2 * math.pi * t * (2 ** -l)) for l in range(num_layers)], dim=1)


def _r16(x: torch.Tensor, on: bool) -> torch.Tensor:
    """Round through fp16 in the forward value only (straight-through): the backward pass stays
    in ``x.dtype`` so the oracle's gradients are not themselves subject to fp16 underflow
    (SURVEY Q15)."""
    if not on:
        return x
    return x + (x.detach().half().to(x.dtype) - x.detach())


def lgp_forward(sd: Dict[str, torch.Tensor], x: torch.Tensor, t: torch.Tensor, *,
                num_layers: int = 9, training: bool = True, emulate_fp16: bool = True,
                update_running: Optional[Dict[str, torch.Tensor]] = None,
                compute_dtype=torch.float32) -> torch.Tensor:
    """x (B,C,h,w), t (B,4,h,w) -> (B*w*h, out) in row order ``(b w h)``.

    ``emulate_fp16``: round at every module boundary the fp16 reference rounds at (the input
    cast, each Linear / BatchNorm output) while accumulating in ``compute_dtype`` - this is how
    an fp16 module with fp32 accumulation behaves.  With it off the function is a smooth
    fp32/fp64 map (used for finite-difference checks)."""
    pos = positional_channels(t, num_layers)
    z = torch.cat((x, t, pos), dim=1)
    B, C, h, w = z.shape
    z = z.permute(0, 3, 2, 1).reshape(B * w * h, C)            # "(b w h) c"
    z = _r16(z.to(compute_dtype), emulate_fp16)
    for i in range(5):
        Wt = sd[f"layers.{LIN[i]}.weight"].to(compute_dtype)
        b = sd[f"layers.{LIN[i]}.bias"].to(compute_dtype)
        z = _r16(z @ Wt.t() + b, emulate_fp16)
        if i == 4:
            break
        z = torch.relu(z)
        bn = BNS[i]
        gamma = sd[f"layers.{bn}.weight"].to(compute_dtype)
        beta = sd[f"layers.{bn}.bias"].to(compute_dtype)
        if training:
            mean = z.mean(dim=0)
            var = z.var(dim=0, unbiased=False)
            if update_running is not None:
                n = z.shape[0]
                rm, rv = update_running[f"layers.{bn}.running_mean"], update_running[f"layers.{bn}.running_var"]
                rm.mul_(0.9).add_(0.1 * mean.detach().to(rm.dtype))
                rv.mul_(0.9).add_(0.1 * (var.detach() * n / (n - 1)).to(rv.dtype))
                update_running[f"layers.{bn}.num_batches_tracked"] += 1
        else:
            mean = sd[f"layers.{bn}.running_mean"].to(compute_dtype)
            var = sd[f"layers.{bn}.running_var"].to(compute_dtype)
        z = _r16((z - mean) * torch.rsqrt(var + 1e-5) * gamma + beta, emulate_fp16)
    return z

"""Oracle: GLIGEN-style injected attention (the reference's SatMixin / AttnModule).  TEST INFRASTRUCTURE.

Op order follows the reference's own files:
  * CLIP-token variant      modules/clip_guided_attn.py:111-125
        s = sketch_proj(sketch_state)                      Linear(1024, C)
        z = sketch_norm(cat([h, s], dim=1))                LayerNorm over N+257 tokens
        a = sketch_attn(z)                                 self-attention (no qkv bias), to_out.0 + bias
        a = a[:, :N, :C];  h = h + scale * Conv1d_1x1(a^T)^T
  * UNet-feature variant    modules/sketch_guided_attn.py:120-132
        z = sketch_norm(h);  a = sketch_attn(z, encoder_hidden_states=res_sample)
        same slice / conv / scale / residual
  * res-sample routing      modules/sketch_guided_attn.py:29-40, 81-82
  * module naming           modules/clip_guided_attn.py:14-27  ("sketch_attn_" + path with '.'->'_')
The attention arithmetic itself (diffusers CrossAttention) is third-party and absent here:
PARITY UNPINNED for arithmetic, op order pinned by reading the files above.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Dict, List, Sequence

import torch
import torch.nn.functional as F

from . import unet as _unet


def module_name(block_path: str) -> str:
    return ("sketch_attn." + block_path).replace(".", "_")


def block_dims(cfg: _unet.UNetConfig) -> List[tuple]:
    """(path, C, heads) for every BasicTransformerBlock in SatMixin.blocks order."""
    boc = cfg.block_out_channels
    rev = tuple(reversed(boc))
    rev_heads = tuple(reversed(cfg.num_heads))
    out = []
    for p in _unet.transformer_block_paths(cfg):
        parts = p.split(".")
        if parts[0] == "down_blocks":
            i = int(parts[1]); out.append((p, boc[i], cfg.num_heads[i]))
        elif parts[0] == "up_blocks":
            i = int(parts[1]); out.append((p, rev[i], rev_heads[i]))
        else:
            out.append((p, boc[-1], cfg.num_heads[-1]))
    return out


def state_dict_manifest(cfg: _unet.UNetConfig, variant: str) -> "OrderedDict[str, tuple]":
    """SatMixin.state_dict() keys/shapes.  variant: 'clip' (has sketch_proj) or 'sketch'."""
    m: "OrderedDict[str, tuple]" = OrderedDict()
    for p, c, _ in block_dims(cfg):
        n = module_name(p)
        if variant == "clip":
            m[f"{n}.sketch_proj.weight"] = (c, 1024)
            m[f"{n}.sketch_proj.bias"] = (c,)
        m[f"{n}.sketch_norm.weight"] = (c,)
        m[f"{n}.sketch_norm.bias"] = (c,)
        for q in ("to_q", "to_k", "to_v"):
            m[f"{n}.sketch_attn.{q}.weight"] = (c, c)
        m[f"{n}.sketch_attn.to_out.0.weight"] = (c, c)
        m[f"{n}.sketch_attn.to_out.0.bias"] = (c,)
        m[f"{n}.sketch_conv.weight"] = (c, c, 1)
        m[f"{n}.sketch_conv.bias"] = (c,)
    return m


def init_state_dict(cfg, variant: str, seed: int = 20260930) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    man = state_dict_manifest(cfg, variant)
    sd = {}
    for k, shp in man.items():
        if ".sketch_norm." in k:
            v = (torch.ones(shp) if k.endswith("weight") else torch.zeros(shp)) \
                + 0.1 * (torch.rand(shp, generator=g) - 0.5)
        else:
            wk = k[: -len("bias")] + "weight" if k.endswith("bias") else k
            fan_in = math.prod(man[wk][1:])
            v = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(fan_in)
        sd[k] = v.half().float()
    return sd


def route_res_samples(res_samples: Sequence[Sequence[torch.Tensor]]) -> List[torch.Tensor]:
    """modules/sketch_guided_attn.py:29-40: per-block K/V sources in SatMixin.blocks order."""
    down, up = (), ()
    mid = (res_samples[-1][-1],)
    for layers in res_samples:
        if len(layers) == 3:
            down += (layers[0], layers[1])
            up += (layers[0], layers[1], layers[1])
    return list(down + up[::-1] + mid)


_r = lambda x: _unet._r(x)      # fp16-storage emulation (oracle/unet.py fp16_storage), identity by default


def _attn_module(sd, n, x, ctx, heads):
    q = _r(F.linear(x, sd[f"{n}.sketch_attn.to_q.weight"]))
    k = _r(F.linear(ctx, sd[f"{n}.sketch_attn.to_k.weight"]))
    v = _r(F.linear(ctx, sd[f"{n}.sketch_attn.to_v.weight"]))
    o = _unet.attention(q, k, v, heads)
    return _r(F.linear(o, sd[f"{n}.sketch_attn.to_out.0.weight"], sd[f"{n}.sketch_attn.to_out.0.bias"]))


def _conv_scale_residual(sd, n, a, h, scale):
    a = a[:, : h.shape[1], : h.shape[2]].permute(0, 2, 1)
    a = scale * F.conv1d(a, sd[f"{n}.sketch_conv.weight"], sd[f"{n}.sketch_conv.bias"])
    return _r(a.permute(0, 2, 1) + h)


def make_clip_inject(sd: Dict[str, torch.Tensor], sketch_state: torch.Tensor, scale: float = 1.0):
    """sketch_state (2,257,1024) = [zeros; clip_hidden] (modules/clip_guided_inf.py:107)."""
    def inject(path, h, heads):
        n = module_name(path)
        c = h.shape[-1]
        s = _r(F.linear(sketch_state.to(h.dtype), sd[f"{n}.sketch_proj.weight"], sd[f"{n}.sketch_proj.bias"]))
        z = _r(F.layer_norm(torch.cat([h, s], dim=1), (c,), sd[f"{n}.sketch_norm.weight"],
                            sd[f"{n}.sketch_norm.bias"], 1e-5))
        a = _attn_module(sd, n, z, z, heads)
        return _conv_scale_residual(sd, n, a, h, scale)
    return inject


def make_sketch_inject(cfg, sd: Dict[str, torch.Tensor], res_samples, scale: float = 1.0):
    routed = route_res_samples(res_samples)
    paths = _unet.transformer_block_paths(cfg)
    table = {p: r.flatten(2).transpose(1, 2) for p, r in zip(paths, routed)}   # b c h w -> b (h w) c

    def inject(path, h, heads):
        n = module_name(path)
        c = h.shape[-1]
        z = _r(F.layer_norm(h, (c,), sd[f"{n}.sketch_norm.weight"], sd[f"{n}.sketch_norm.bias"], 1e-5))
        a = _attn_module(sd, n, z, table[path].to(h.dtype), heads)
        return _conv_scale_residual(sd, n, a, h, scale)
    return inject

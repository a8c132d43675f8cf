"""Oracle: CLIP vision transformer (transformers CLIPVisionModel) -> last_hidden_state.  TEST INFRASTRUCTURE.

Restates what the reference reaches at ``modules/clip_guided_inf.py:100-104``:
    sketch_encoder = CLIPVisionModel.from_pretrained("openai/clip-vit-large-patch14")            (:49-51)
    h = sketch_encoder(pixel_values, output_hidden_states=True).last_hidden_state                (:103) -> (B, 257, 1024)
    sat_model.set_state(torch.stack([zeros_like(h), h]).squeeze(1))                              (:105)
The arithmetic is third-party ``transformers`` (unpinned in the reference; call sites clip_guided_inf.py:12,49-54).
Unlike diffusers, transformers IS importable in the build container (5.15.0), so this restatement is PINNED:
``tools/gen_golden_clip.py`` runs transformers' own CLIPVisionModel on a seeded small configuration and commits
weights / input / output as ``tests/golden/clip_vision_tiny.npz``; ``tests/test_oracle.py`` checks this file
against it (and against a live transformers model when the package is importable).

Model (ViT, pre-LN):  patch_embedding Conv2d(3, D, P, stride P, no bias) -> tokens [cls ; patches] + position_embedding
-> pre_layrnorm (sic) -> L x { x + out_proj(softmax(q k^T / sqrt(d)) v) with q,k,v = Linear(LN1(x)) (+bias);
x + fc2(quick_gelu(fc1(LN2(x)))) } ; quick_gelu(x) = x * sigmoid(1.702 x); LayerNorm eps 1e-5.
``last_hidden_state`` is the encoder output WITHOUT post_layernorm (that one only feeds the pooled output).
State-dict keys are transformers' (an optional ``vision_model.`` prefix, used by transformers 4.x, is accepted).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class CLIPVisionConfig:
    hidden_size: int = 1024
    intermediate_size: int = 4096
    num_hidden_layers: int = 24
    num_attention_heads: int = 16
    image_size: int = 224
    patch_size: int = 14
    layer_norm_eps: float = 1e-5

    @property
    def num_tokens(self) -> int:
        return (self.image_size // self.patch_size) ** 2 + 1


VIT_L_14 = CLIPVisionConfig()
TINY_CLIP = CLIPVisionConfig(hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                             image_size=56, patch_size=14)


def param_shapes(cfg: CLIPVisionConfig) -> "OrderedDict[str, tuple]":
    D, I = cfg.hidden_size, cfg.intermediate_size
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["embeddings.class_embedding"] = (D,)
    s["embeddings.patch_embedding.weight"] = (D, 3, cfg.patch_size, cfg.patch_size)
    s["embeddings.position_embedding.weight"] = (cfg.num_tokens, D)
    s["pre_layrnorm.weight"] = (D,)
    s["pre_layrnorm.bias"] = (D,)
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{l}"
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[f"{p}.self_attn.{n}.weight"] = (D, D)
            s[f"{p}.self_attn.{n}.bias"] = (D,)
        s[f"{p}.layer_norm1.weight"] = (D,)
        s[f"{p}.layer_norm1.bias"] = (D,)
        s[f"{p}.mlp.fc1.weight"] = (I, D)
        s[f"{p}.mlp.fc1.bias"] = (I,)
        s[f"{p}.mlp.fc2.weight"] = (D, I)
        s[f"{p}.mlp.fc2.bias"] = (D,)
        s[f"{p}.layer_norm2.weight"] = (D,)
        s[f"{p}.layer_norm2.bias"] = (D,)
    s["post_layernorm.weight"] = (D,)
    s["post_layernorm.bias"] = (D,)
    return s


def init_weights(cfg: CLIPVisionConfig, seed: int = 20261002) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights (fp16-representable): Linear / conv U(+-1/sqrt(fan_in)), LN gamma 1 + 0.1 N,
    biases / class embedding 0.05 N, position embedding 0.02 N."""
    g = torch.Generator().manual_seed(seed)
    W: Dict[str, torch.Tensor] = {}
    for k, shp in param_shapes(cfg).items():
        if ("norm" in k) and k.endswith(".weight"):
            w = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias") or k.endswith("class_embedding"):
            w = 0.05 * torch.randn(shp, generator=g)
        elif "position_embedding" in k:
            w = 0.02 * torch.randn(shp, generator=g)
        else:
            w = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(math.prod(shp[1:]))
        W[k] = w.half().float()
    return W


def strip_prefix(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {(k[len("vision_model."):] if k.startswith("vision_model.") else k): v for k, v in sd.items()}


def quick_gelu(x: torch.Tensor) -> torch.Tensor:
    return x * torch.sigmoid(1.702 * x)


def last_hidden_state(cfg: CLIPVisionConfig, W: Dict[str, torch.Tensor], pixel_values: torch.Tensor) -> torch.Tensor:
    """pixel_values (B, 3, S, S) -> (B, 1 + (S/P)^2, D)."""
    B = pixel_values.shape[0]
    D, H = cfg.hidden_size, cfg.num_attention_heads
    d = D // H
    eps = cfg.layer_norm_eps
    x = F.conv2d(pixel_values, W["embeddings.patch_embedding.weight"], stride=cfg.patch_size)      # (B, D, g, g)
    x = x.flatten(2).transpose(1, 2)                                                               # (B, g*g, D)
    cls = W["embeddings.class_embedding"].expand(B, 1, D)
    x = torch.cat([cls, x], dim=1) + W["embeddings.position_embedding.weight"][None]
    x = F.layer_norm(x, (D,), W["pre_layrnorm.weight"], W["pre_layrnorm.bias"], eps)
    N = x.shape[1]
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{l}"
        h = F.layer_norm(x, (D,), W[p + ".layer_norm1.weight"], W[p + ".layer_norm1.bias"], eps)
        q = F.linear(h, W[p + ".self_attn.q_proj.weight"], W[p + ".self_attn.q_proj.bias"])
        k = F.linear(h, W[p + ".self_attn.k_proj.weight"], W[p + ".self_attn.k_proj.bias"])
        v = F.linear(h, W[p + ".self_attn.v_proj.weight"], W[p + ".self_attn.v_proj.bias"])
        q, k, v = (t.reshape(B, N, H, d).transpose(1, 2) for t in (q, k, v))
        a = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, N, D)
        x = x + F.linear(a, W[p + ".self_attn.out_proj.weight"], W[p + ".self_attn.out_proj.bias"])
        h = F.layer_norm(x, (D,), W[p + ".layer_norm2.weight"], W[p + ".layer_norm2.bias"], eps)
        h = quick_gelu(F.linear(h, W[p + ".mlp.fc1.weight"], W[p + ".mlp.fc1.bias"]))
        x = x + F.linear(h, W[p + ".mlp.fc2.weight"], W[p + ".mlp.fc2.bias"])
    return x

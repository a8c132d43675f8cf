"""Oracle: CLIP text transformer (transformers CLIPTextModel) -> last_hidden_state.  TEST INFRASTRUCTURE.

Restates what the reference reaches at ``modules/pipeline.py:55-57``:
    text_embeddings = self._encode_prompt(prompt, device, num_images_per_prompt, do_cfg, negative_prompt)
which in diffusers 0.12 is ``self.text_encoder(text_input_ids)[0]`` = CLIPTextModel(...).last_hidden_state for the
negative and the positive prompts, concatenated [uncond ; cond].  The arithmetic is third-party ``transformers``
(unpinned in the reference).  transformers IS importable in the build container, so this restatement is PINNED:
``tools/gen_golden_clip_text.py`` runs transformers' own CLIPTextModel on a seeded small configuration and commits
weights / input ids / output as ``tests/golden/clip_text_tiny.npz``; ``tests/test_oracle.py`` checks this file against
it (and against a live transformers model when the package is importable).

Model (pre-LN, causal): token_embedding[ids] + position_embedding -> L x { x + out_proj(softmax(q k^T / sqrt(d) + causal
mask) v), q,k,v = Linear(LN1(x)) (+bias);  x + fc2(act(fc1(LN2(x)))) } -> final_layer_norm.  act = quick_gelu
(x sigmoid(1.702 x), SD 1.x) or exact gelu (SD 2.x).  ``last_hidden_state`` INCLUDES final_layer_norm (unlike the
vision tower's post_layernorm, which only feeds the pooled output).  An optional ``text_model.`` key prefix
(transformers 4.x) is accepted.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class CLIPTextConfig:
    vocab_size: int = 49408
    hidden_size: int = 768
    intermediate_size: int = 3072
    num_hidden_layers: int = 12
    num_attention_heads: int = 12
    max_position_embeddings: int = 77
    hidden_act: str = "quick_gelu"
    layer_norm_eps: float = 1e-5


SD15_TEXT = CLIPTextConfig()
SD21_TEXT = CLIPTextConfig(hidden_size=1024, intermediate_size=4096, num_hidden_layers=23, num_attention_heads=16,
                           hidden_act="gelu")
TINY_TEXT = CLIPTextConfig(vocab_size=1000, hidden_size=64, intermediate_size=128, num_hidden_layers=2,
                           num_attention_heads=4)


def param_shapes(cfg: CLIPTextConfig) -> "OrderedDict[str, tuple]":
    D, I = cfg.hidden_size, cfg.intermediate_size
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["embeddings.token_embedding.weight"] = (cfg.vocab_size, D)
    s["embeddings.position_embedding.weight"] = (cfg.max_position_embeddings, D)
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
    s["final_layer_norm.weight"] = (D,)
    s["final_layer_norm.bias"] = (D,)
    return s


def num_params(cfg: CLIPTextConfig) -> int:
    return sum(math.prod(v) for v in param_shapes(cfg).values())


def init_weights(cfg: CLIPTextConfig, seed: int = 20261003) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights (fp16-representable): Linear U(+-1/sqrt(fan_in)), LN gamma 1 + 0.1 N, biases 0.05 N,
    token embedding 0.5 N, position embedding 0.1 N."""
    g = torch.Generator().manual_seed(seed)
    W: Dict[str, torch.Tensor] = {}
    for k, shp in param_shapes(cfg).items():
        if ("norm" in k) and k.endswith(".weight"):
            w = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            w = 0.05 * torch.randn(shp, generator=g)
        elif "token_embedding" in k:
            w = 0.5 * torch.randn(shp, generator=g)
        elif "position_embedding" in k:
            w = 0.1 * torch.randn(shp, generator=g)
        else:
            w = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(math.prod(shp[1:]))
        W[k] = w.half().float()
    return W


def strip_prefix(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    return {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in sd.items()}


def activation(name: str, x: torch.Tensor) -> torch.Tensor:
    if name == "quick_gelu":
        return x * torch.sigmoid(1.702 * x)
    if name == "gelu":
        return 0.5 * x * (1.0 + torch.erf(x / math.sqrt(2.0)))
    raise ValueError(f"unsupported hidden_act {name!r}")


def last_hidden_state(cfg: CLIPTextConfig, W: Dict[str, torch.Tensor], input_ids: torch.Tensor) -> torch.Tensor:
    """input_ids (B, L <= max_position_embeddings) int64 -> (B, L, D)."""
    B, L = input_ids.shape
    D, H = cfg.hidden_size, cfg.num_attention_heads
    d = D // H
    eps = cfg.layer_norm_eps
    x = W["embeddings.token_embedding.weight"][input_ids] + W["embeddings.position_embedding.weight"][:L][None]
    mask = torch.full((L, L), float("-inf")).triu(1)            # key j > query i is hidden
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{l}"
        h = F.layer_norm(x, (D,), W[p + ".layer_norm1.weight"], W[p + ".layer_norm1.bias"], eps)
        q = F.linear(h, W[p + ".self_attn.q_proj.weight"], W[p + ".self_attn.q_proj.bias"])
        k = F.linear(h, W[p + ".self_attn.k_proj.weight"], W[p + ".self_attn.k_proj.bias"])
        v = F.linear(h, W[p + ".self_attn.v_proj.weight"], W[p + ".self_attn.v_proj.bias"])
        q, k, v = (t.reshape(B, L, H, d).transpose(1, 2) for t in (q, k, v))
        a = torch.softmax(q @ k.transpose(-1, -2) * d ** -0.5 + mask, dim=-1) @ v
        a = a.transpose(1, 2).reshape(B, L, D)
        x = x + F.linear(a, W[p + ".self_attn.out_proj.weight"], W[p + ".self_attn.out_proj.bias"])
        h = F.layer_norm(x, (D,), W[p + ".layer_norm2.weight"], W[p + ".layer_norm2.bias"], eps)
        h = activation(cfg.hidden_act, F.linear(h, W[p + ".mlp.fc1.weight"], W[p + ".mlp.fc1.bias"]))
        x = x + F.linear(h, W[p + ".mlp.fc2.weight"], W[p + ".mlp.fc2.bias"])
    return F.layer_norm(x, (D,), W["final_layer_norm.weight"], W["final_layer_norm.bias"], eps)

"""CLIP vision tower (ViT) on the libskg.so kernels: the producer of the sketch tokens for the CLIP-guided variant.

Replaces ``CLIPVisionModel(...)(pixel_values, output_hidden_states=True).last_hidden_state`` at
modules/clip_guided_inf.py:49-54,103 (third-party transformers; restated and PINNED against transformers' own class in
oracle/clip_vision.py + tests/golden/clip_vision_tiny.npz).  Runs once per image; SURVEY.md section 8f rank 3.

Layout: tokens fp16 [B * Lp, D] with every image's 1 + g*g tokens padded to Lp = a multiple of 8 rows (257 -> 264) so
the transposed V panel keeps 16-byte aligned rows; pad rows are never read as keys (Nkv = 257) and are dropped at the
end.  Patch embedding = GEMM over the unfolded 14x14x3 patches (K = 588 zero-padded to 608), q/k/v fused into one GEMM
(with bias), flash attention (d = 64), out-proj / fc2 GEMMs with the residual in the epilogue, quick_gelu and LayerNorm
kernels.  The image pre-processing (CLIPImageProcessor: resize, crop, normalise) stays with the caller.
"""
from __future__ import annotations

import os
from typing import Dict, Optional

import torch

from . import ops
from .config import CLIPVisionConfig, VIT_L_14
from .unet import _h


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def strip_prefix(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """transformers 4.x prefixes the tower's keys with ``vision_model.``; 5.x does not.  Both load."""
    return {(k[len("vision_model."):] if k.startswith("vision_model.") else k): v for k, v in sd.items()}


class HipCLIPVision:
    def __init__(self, cfg: CLIPVisionConfig, state_dict: Dict[str, torch.Tensor], device="cuda"):
        self.cfg, self.dev = cfg, torch.device(device)
        self.Lp = _round_up(cfg.num_tokens, 8)
        self.Kp = _round_up(3 * cfg.patch_size ** 2, 32)
        self.W = self._pack(strip_prefix(state_dict))
        self._pos_rep: Dict[int, torch.Tensor] = {}

    def _pack(self, sd):
        cfg, dev = self.cfg, self.dev
        W: Dict[str, torch.Tensor] = {}
        D = cfg.hidden_size
        pe = sd["embeddings.patch_embedding.weight"].detach().float().reshape(D, -1)         # [D, 3*P*P] (c, y, x)
        W["patch"] = _h(torch.nn.functional.pad(pe, (0, self.Kp - pe.shape[1])), dev)
        pos = torch.zeros(self.Lp, D)
        pos[:cfg.num_tokens] = sd["embeddings.position_embedding.weight"].detach().float()
        pos[0] += sd["embeddings.class_embedding"].detach().float()                           # cls token + its position
        W["pos"] = _h(pos, dev)
        for k in ("pre_layrnorm.weight", "pre_layrnorm.bias"):
            W[k] = _h(sd[k].detach().float(), dev)
        for l in range(cfg.num_hidden_layers):
            p = f"encoder.layers.{l}"
            W[p + ".qkv.weight"] = _h(torch.cat([sd[f"{p}.self_attn.{n}.weight"].detach().float()
                                                 for n in ("q_proj", "k_proj", "v_proj")]), dev)
            W[p + ".qkv.bias"] = _h(torch.cat([sd[f"{p}.self_attn.{n}.bias"].detach().float()
                                               for n in ("q_proj", "k_proj", "v_proj")]), dev)
            for n in ("self_attn.out_proj", "mlp.fc1", "mlp.fc2", "layer_norm1", "layer_norm2"):
                W[f"{p}.{n}.weight"] = _h(sd[f"{p}.{n}.weight"].detach().float(), dev)
                W[f"{p}.{n}.bias"] = _h(sd[f"{p}.{n}.bias"].detach().float(), dev)
        return W

    def to(self, device):
        if torch.device(device) != self.dev:
            self.dev = torch.device(device)
            self.W = {k: v.to(self.dev) for k, v in self.W.items()}
            self._pos_rep = {}
        return self

    @torch.no_grad()
    def last_hidden_state(self, pixel_values: torch.Tensor) -> torch.Tensor:
        """pixel_values [B, 3, S, S] (already CLIP-normalised) -> fp16 [B, 1 + (S/P)^2, D]."""
        cfg, W = self.cfg, self.W
        B, _, S, S2 = pixel_values.shape
        P, D, H = cfg.patch_size, cfg.hidden_size, cfg.num_attention_heads
        if S != cfg.image_size or S2 != S:
            raise ValueError(f"CLIP vision tower expects {cfg.image_size}x{cfg.image_size} images, got {S}x{S2}")
        g, N, Lp, d = S // P, cfg.num_tokens, self.Lp, D // H
        # unfold the non-overlapping patches: pure data movement (no arithmetic) done with torch views
        x = pixel_values.to(self.dev, torch.float32).reshape(B, 3, g, P, g, P).permute(0, 2, 4, 1, 3, 5)
        cols = torch.zeros(B * g * g, self.Kp, device=self.dev, dtype=torch.float16)
        cols[:, :3 * P * P] = x.reshape(B * g * g, 3 * P * P)
        tok = torch.zeros(B, Lp, D, device=self.dev, dtype=torch.float16)
        emb = ops.gemm(cols, W["patch"])                                       # [B*g*g, D]
        tok[:, 1:N] = emb.view(B, g * g, D)
        if B not in self._pos_rep:
            self._pos_rep[B] = W["pos"].repeat(B, 1).contiguous()
        x = ops.axpby(tok.view(B * Lp, D), self._pos_rep[B])                   # + position (+ class) embedding
        x = ops.layernorm(x, W["pre_layrnorm.weight"], W["pre_layrnorm.bias"], cfg.layer_norm_eps)
        scale = d ** -0.5
        for l in range(cfg.num_hidden_layers):
            p = f"encoder.layers.{l}"
            h = ops.layernorm(x, W[p + ".layer_norm1.weight"], W[p + ".layer_norm1.bias"], cfg.layer_norm_eps)
            qkv = ops.gemm(h, W[p + ".qkv.weight"], bias=W[p + ".qkv.bias"])
            a = ops.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], qkv[:, 2 * D:], B, H, Lp, N, Lp, d, scale, v_rows=True)
            x = ops.gemm(a, W[p + ".self_attn.out_proj.weight"], bias=W[p + ".self_attn.out_proj.bias"], residual=x)
            h = ops.layernorm(x, W[p + ".layer_norm2.weight"], W[p + ".layer_norm2.bias"], cfg.layer_norm_eps)
            f = ops.gemm(h, W[p + ".mlp.fc1.weight"], bias=W[p + ".mlp.fc1.bias"])
            ops.quick_gelu(f, out=f)
            x = ops.gemm(f, W[p + ".mlp.fc2.weight"], bias=W[p + ".mlp.fc2.bias"], residual=x)
        return x.view(B, Lp, D)[:, :N].contiguous()


class _VisionOutput:
    def __init__(self, last_hidden_state):
        self.last_hidden_state = last_hidden_state


class CLIPVisionModel:
    """Facade with the surface modules/clip_guided_inf.py:49-54,103 uses of transformers.CLIPVisionModel:
    ``from_pretrained(path)``, ``load_state_dict(sd)``, ``.to(device, dtype=)``, ``.device`` / ``.dtype``,
    ``model(pixel_values, output_hidden_states=True).last_hidden_state``."""

    def __init__(self, cfg: CLIPVisionConfig = VIT_L_14, state_dict: Optional[Dict[str, torch.Tensor]] = None):
        from . import synthetic
        self.cfg = self.config = cfg
        self._sd = strip_prefix(state_dict) if state_dict is not None else synthetic.clip_vision_state_dict(cfg)
        self._hip: Optional[HipCLIPVision] = None
        self.device, self.dtype = torch.device("cpu"), torch.float16

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, config: Optional[CLIPVisionConfig] = None, **kwargs):
        sd = None
        if pretrained_model_name_or_path and os.path.isdir(pretrained_model_name_or_path):
            st = os.path.join(pretrained_model_name_or_path, "model.safetensors")
            pt = os.path.join(pretrained_model_name_or_path, "pytorch_model.bin")
            if os.path.exists(st):
                from safetensors.torch import load_file
                sd = load_file(st)
            elif os.path.exists(pt):
                sd = torch.load(pt, map_location="cpu")
            if sd is not None:
                sd = {k: v for k, v in sd.items() if k.startswith(("vision_model.", "embeddings.", "encoder.",
                                                                   "pre_layrnorm.", "post_layernorm."))}
        return cls(config or VIT_L_14, sd)

    def state_dict(self):
        return self._sd

    def load_state_dict(self, sd, strict: bool = True):
        sd = strip_prefix(sd)
        missing = [k for k in self._sd if k not in sd and "position_ids" not in k]
        if strict and missing:
            raise RuntimeError(f"CLIPVisionModel.load_state_dict: missing keys {missing[:4]} ...")
        self._sd = {k: v for k, v in sd.items() if "position_ids" not in k}
        if self._hip is not None:
            self._hip = HipCLIPVision(self.cfg, self._sd, self.device)
        return self

    def to(self, device=None, dtype=None):
        if isinstance(device, torch.dtype):
            device, dtype = None, device
        if device is not None:
            self.device = torch.device(device)
            if self.device.type == "cuda":
                if self._hip is None:
                    self._hip = HipCLIPVision(self.cfg, self._sd, self.device)
                else:
                    self._hip.to(self.device)
        return self

    def eval(self):
        return self

    def __call__(self, pixel_values, output_hidden_states: bool = False, **kwargs):
        if self._hip is None:
            raise RuntimeError("CLIPVisionModel: call .to('cuda') first - the tower runs on libskg.so kernels only")
        return _VisionOutput(self._hip.last_hidden_state(pixel_values))

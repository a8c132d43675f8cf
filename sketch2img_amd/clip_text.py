"""CLIP text encoder on the libskg.so kernels: prompts -> the (B, 77, D) conditioning the UNet cross-attends to.

Replaces ``self.text_encoder(text_input_ids)[0]`` inside StableDiffusionPipeline._encode_prompt, which the reference
calls at modules/pipeline.py:55-57 (third-party transformers CLIPTextModel; restated and PINNED against transformers'
own class in oracle/clip_text.py + tests/golden/clip_text_tiny.npz).  Runs once per prompt batch.

Layout: tokens fp16 [B * Lp, D], every prompt's L (= 77) tokens padded to Lp = a multiple of 8 rows (80) so the
transposed V panel keeps 16-byte aligned rows; pad rows are never read as keys (Nkv = L) and are dropped at the end.
The token-embedding row gather is data movement (torch.index_select on the fp16 table); q/k/v are one GEMM with bias,
the causal softmax(QK^T)V is the flash kernel's CAUSAL instantiation (skg_attn_fwd_causal), out-proj / fc2 carry the
residual in the GEMM epilogue, quick_gelu (SD 1.x) or exact gelu (SD 2.x) and LayerNorm are their own kernels.
Tokenisation (BPE) is host-side text processing and stays with transformers' CLIPTokenizer (`PromptEncoder`).
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional

import torch

from . import ops
from .config import CLIPTextConfig, SD15_TEXT, SD21_TEXT
from .unet import _h


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


def strip_prefix(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
    """transformers 4.x prefixes the keys with ``text_model.``; 5.x does not.  Both load."""
    return {(k[len("text_model."):] if k.startswith("text_model.") else k): v for k, v in sd.items()}


class HipCLIPText:
    def __init__(self, cfg: CLIPTextConfig, state_dict: Dict[str, torch.Tensor], device="cuda"):
        if cfg.hidden_act not in ("quick_gelu", "gelu"):
            raise ValueError(f"CLIP text encoder: unsupported hidden_act {cfg.hidden_act!r}")
        self.cfg, self.dev = cfg, torch.device(device)
        self.W = self._pack(strip_prefix(state_dict))

    def _pack(self, sd):
        cfg, dev = self.cfg, self.dev
        W: Dict[str, torch.Tensor] = {}
        W["tok"] = _h(sd["embeddings.token_embedding.weight"].detach().float(), dev)
        W["pos"] = _h(sd["embeddings.position_embedding.weight"].detach().float(), dev)
        for k in ("final_layer_norm.weight", "final_layer_norm.bias"):
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
        return self

    @torch.no_grad()
    def last_hidden_state(self, input_ids: torch.Tensor) -> torch.Tensor:
        """input_ids int [B, L <= max_position_embeddings] -> fp16 [B, L, D] (final_layer_norm applied)."""
        cfg, W = self.cfg, self.W
        B, L = input_ids.shape
        if L > cfg.max_position_embeddings:
            raise ValueError(f"CLIP text encoder: {L} tokens > max_position_embeddings {cfg.max_position_embeddings}")
        ids = input_ids.to(self.dev, torch.long)
        if int(ids.min()) < 0 or int(ids.max()) >= cfg.vocab_size:
            raise ValueError("CLIP text encoder: token id outside the vocabulary")
        D, H = cfg.hidden_size, cfg.num_attention_heads
        d, Lp = D // H, _round_up(L, 8)
        tok = torch.zeros(B, Lp, D, device=self.dev, dtype=torch.float16)
        pos = torch.zeros(B, Lp, D, device=self.dev, dtype=torch.float16)
        tok[:, :L] = W["tok"].index_select(0, ids.reshape(-1)).view(B, L, D)       # row gather: data movement only
        pos[:, :L] = W["pos"][:L]
        x = ops.axpby(tok.view(B * Lp, D), pos.view(B * Lp, D))
        act = ops.quick_gelu if cfg.hidden_act == "quick_gelu" else ops.gelu
        scale = d ** -0.5
        for l in range(cfg.num_hidden_layers):
            p = f"encoder.layers.{l}"
            h = ops.layernorm(x, W[p + ".layer_norm1.weight"], W[p + ".layer_norm1.bias"], cfg.layer_norm_eps)
            qkv = ops.gemm(h, W[p + ".qkv.weight"], bias=W[p + ".qkv.bias"])
            vt = ops.transpose(qkv[:, 2 * D:])
            a = ops.attn_fwd(qkv[:, :D], qkv[:, D:2 * D], vt, B, H, Lp, L, Lp, d, scale, causal=True)
            x = ops.gemm(a, W[p + ".self_attn.out_proj.weight"], bias=W[p + ".self_attn.out_proj.bias"], residual=x)
            h = ops.layernorm(x, W[p + ".layer_norm2.weight"], W[p + ".layer_norm2.bias"], cfg.layer_norm_eps)
            f = ops.gemm(h, W[p + ".mlp.fc1.weight"], bias=W[p + ".mlp.fc1.bias"])
            act(f, out=f)
            x = ops.gemm(f, W[p + ".mlp.fc2.weight"], bias=W[p + ".mlp.fc2.bias"], residual=x)
        x = ops.layernorm(x, W["final_layer_norm.weight"], W["final_layer_norm.bias"], cfg.layer_norm_eps)
        return x.view(B, Lp, D)[:, :L].contiguous()


class _TextOutput(tuple):
    """``out[0]`` and ``out.last_hidden_state`` both work, like transformers' BaseModelOutputWithPooling."""

    def __new__(cls, last_hidden_state):
        self = super().__new__(cls, (last_hidden_state,))
        self.last_hidden_state = last_hidden_state
        return self


def _config_from_folder(path: Optional[str]) -> CLIPTextConfig:
    import json
    if path and os.path.exists(os.path.join(path, "config.json")):
        with open(os.path.join(path, "config.json")) as f:
            c = json.load(f)
        c = c.get("text_config", c) if "hidden_size" not in c else c
        keys = CLIPTextConfig.__dataclass_fields__.keys()
        return CLIPTextConfig(**{k: c[k] for k in keys if k in c})
    return SD15_TEXT


class CLIPTextModel:
    """Facade with the surface diffusers' _encode_prompt uses of transformers.CLIPTextModel: ``from_pretrained(path)``,
    ``load_state_dict(sd)``, ``.to(device)``, ``.device`` / ``.dtype`` / ``.config``, ``model(input_ids)[0]``."""

    def __init__(self, cfg: CLIPTextConfig = SD15_TEXT, state_dict: Optional[Dict[str, torch.Tensor]] = None):
        from . import synthetic
        self.cfg = self.config = cfg
        self._sd = strip_prefix(state_dict) if state_dict is not None else synthetic.clip_text_state_dict(cfg)
        self._hip: Optional[HipCLIPText] = None
        self.device, self.dtype = torch.device("cpu"), torch.float16

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, config: Optional[CLIPTextConfig] = None,
                        subfolder: Optional[str] = None, **kwargs):
        path = pretrained_model_name_or_path
        if path and subfolder:
            path = os.path.join(path, subfolder)
        sd = None
        if path and os.path.isdir(path):
            st, pt = os.path.join(path, "model.safetensors"), os.path.join(path, "pytorch_model.bin")
            if os.path.exists(st):
                from safetensors.torch import load_file
                sd = load_file(st)
            elif os.path.exists(pt):
                sd = torch.load(pt, map_location="cpu")
            if sd is not None:
                sd = {k: v for k, v in sd.items() if "position_ids" not in k}
        return cls(config or _config_from_folder(path), sd)

    def state_dict(self):
        return self._sd

    def load_state_dict(self, sd, strict: bool = True):
        sd = strip_prefix(sd)
        missing = [k for k in self._sd if k not in sd and "position_ids" not in k]
        if strict and missing:
            raise RuntimeError(f"CLIPTextModel.load_state_dict: missing keys {missing[:4]} ...")
        self._sd = {k: v for k, v in sd.items() if "position_ids" not in k}
        if self._hip is not None:
            self._hip = HipCLIPText(self.cfg, self._sd, self.device)
        return self

    def to(self, device=None, dtype=None):
        if isinstance(device, torch.dtype):
            device, dtype = None, device
        if device is not None:
            self.device = torch.device(device)
            if self.device.type == "cuda":
                if self._hip is None:
                    self._hip = HipCLIPText(self.cfg, self._sd, self.device)
                else:
                    self._hip.to(self.device)
        return self

    def eval(self):
        return self

    def __call__(self, input_ids, attention_mask=None, **kwargs):
        if self._hip is None:
            raise RuntimeError("CLIPTextModel: call .to('cuda') first - the encoder runs on libskg.so kernels only")
        return _TextOutput(self._hip.last_hidden_state(input_ids))


class PromptEncoder:
    """``(list[str]) -> (B, 77, D)``: the tokenizer + text-encoder half of diffusers' _encode_prompt
    (modules/pipeline.py:55-57): pad to ``model_max_length`` with truncation, then ``text_encoder(ids)[0]``.
    ``tokenizer`` is any callable with transformers' tokenizer call convention (CLIPTokenizer in practice)."""

    def __init__(self, tokenizer, text_model: CLIPTextModel):
        self.tokenizer, self.text_model = tokenizer, text_model

    @classmethod
    def from_pretrained(cls, folder: str):
        """``folder`` is a diffusers checkpoint directory with ``tokenizer/`` and ``text_encoder/`` inside."""
        from transformers import CLIPTokenizer          # BPE tables + host-side string processing only
        tok = CLIPTokenizer.from_pretrained(os.path.join(folder, "tokenizer"))
        return cls(tok, CLIPTextModel.from_pretrained(folder, subfolder="text_encoder"))

    def to(self, device):
        self.text_model.to(device)
        return self

    def __call__(self, prompts: List[str]) -> torch.Tensor:
        L = getattr(self.tokenizer, "model_max_length", None) or self.text_model.cfg.max_position_embeddings
        L = min(L, self.text_model.cfg.max_position_embeddings)
        enc = self.tokenizer(list(prompts), padding="max_length", max_length=L, truncation=True, return_tensors="pt")
        return self.text_model(enc["input_ids"])[0]

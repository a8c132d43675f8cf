"""Shared skeleton of the two SatMixin variants (modules/clip_guided_attn.py, modules/sketch_guided_attn.py)."""
from __future__ import annotations

import torch
import torch.nn as nn

from ..inject import HipInjector, block_dims, module_name


class _AttnParams(nn.Module):
    """Parameter holder with the reference AttnModule's names: sketch_proj (CLIP variant only), sketch_norm,
    sketch_attn.{to_q,to_k,to_v,to_out.0}, sketch_conv (modules/clip_guided_attn.py:52-63)."""

    def __init__(self, name: str, dim: int, heads: int, with_proj: bool):
        super().__init__()
        self.name = name
        self.heads = heads
        if with_proj:
            self.sketch_proj = nn.Linear(1024, dim)
        self.sketch_norm = nn.LayerNorm(dim)
        attn = nn.Module()
        attn.to_q = nn.Linear(dim, dim, bias=False)
        attn.to_k = nn.Linear(dim, dim, bias=False)
        attn.to_v = nn.Linear(dim, dim, bias=False)
        attn.to_out = nn.ModuleList([nn.Linear(dim, dim), nn.Dropout(0.0)])
        self.sketch_attn = attn
        self.sketch_conv = nn.Conv1d(dim, dim, 1)
        self.sketch_scale = 1.0

    def set_scale(self, scale):
        self.sketch_scale = scale


class SatMixinBase(nn.Module):
    variant = "clip"

    def __init__(self, unet):
        super().__init__()
        self._unet = [unet]                       # not registered as a submodule (the reference holds references)
        self.blocks = []
        for path, dim, heads in block_dims(unet.cfg):
            name = module_name(path)
            print(f"Injected: sketch_attn.{path}")
            blk = _AttnParams(name, dim, heads, self.variant == "clip")
            self.blocks.append(blk)
        names = set()
        for blk in self.blocks:
            assert blk.name not in names, f"duplicated module name: {blk.name}"
            names.add(blk.name)
            self.add_module(blk.name, blk)
        self._engine = None
        self._scale = 1.0

    def _injector(self) -> HipInjector:
        unet = self._unet[0]
        key = tuple(int(p._version) for p in self.parameters())
        if self._engine is None or self._engine[0] != key:
            eng = HipInjector(unet.cfg, self.state_dict(), self.variant, unet.device)
            eng.set_scale(self._scale)
            self._engine = (key, eng)
        unet.hip.inject = self._engine[1]
        return self._engine[1]

    def set_scale(self, scale):
        self._scale = float(scale)
        for blk in self.blocks:
            blk.set_scale(scale)
        if self._engine is not None:
            self._engine[1].set_scale(scale)

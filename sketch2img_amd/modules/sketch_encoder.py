"""Drop-in for the reference's modules/sketch_encoder.py: SketchEncoder.

The reference class is a UNet2DConditionModel subclass whose forward stops after the down path and returns
``UNet2DConditionOutput(sample=down_block_res_samples)`` (modules/sketch_encoder.py:39-98): per down block the tuple
of hidden states after each (resnet, attention) pair and after the downsampler.  Here it wraps the HIP UNet
(``HipUNet.forward(..., down_only=True)``) and returns the same structure as NCHW fp16 tensors, ready for
``modules.sketch_guided_attn.SatMixin.set_res_samples``.
"""
from __future__ import annotations

from types import SimpleNamespace

import torch

from .pipeline import UNetFacade


class SketchEncoder(UNetFacade):
    def __call__(self, sample, timestep, encoder_hidden_states, **kwargs):
        from .. import ops
        from ..unet import CIN_PAD
        rows, _, h, w = sample.shape
        assert h == w
        net = self.hip
        if net.ctx is None or net.ctx.get("src") is not encoder_hidden_states:
            net.prepare_context(encoder_hidden_states)
            net.ctx["src"] = encoder_hidden_states
        x32 = ops.nchw_to_nhwc(sample.to(self._device, torch.float32).contiguous(), CIN_PAD)
        blocks = net.forward(x32, int(timestep), rows, h, down_only=True)
        res = [tuple(t.reshape(rows, s, s, -1).permute(0, 3, 1, 2) for t, s in blk) for blk in blocks]
        return SimpleNamespace(sample=res)

    forward = __call__

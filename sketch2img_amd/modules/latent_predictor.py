"""Drop-in for the reference's modules/latent_predictor.py: LatentEdgePredictor and hook_unet.

``LatentEdgePredictor`` keeps the reference's nn.Module skeleton (modules/latent_predictor.py:9-35) so that
its ``state_dict()`` has the same 30 keys / shapes and ``torch.load`` checkpoints (``edge_predictor.pt``,
app.py:67-68) load unchanged; ``forward`` (:37-45) runs on the HIP kernels.  There is no PyTorch fallback.
"""
from __future__ import annotations

import math
from typing import List

import torch
import torch.nn as nn


class LatentEdgePredictor(nn.Module):
    def __init__(self, input_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        self.layers = nn.Sequential(
            nn.Linear(input_dim, 512), nn.ReLU(), nn.BatchNorm1d(num_features=512),
            nn.Linear(512, 256), nn.ReLU(), nn.BatchNorm1d(num_features=256),
            nn.Linear(256, 128), nn.ReLU(), nn.BatchNorm1d(num_features=128),
            nn.Linear(128, 64), nn.ReLU(), nn.BatchNorm1d(num_features=64),
            nn.Linear(64, output_dim),
        )
        for _, module in self.layers.named_modules():          # modules/latent_predictor.py:32-35
            if module.__class__.__name__ == "Linear":
                nn.init.kaiming_uniform_(module.weight)
                nn.init.zeros_(module.bias)
        self._hip = None
        self._hip_key = None

    def _engine(self, tap_channels, device):
        """HipLGP built from the current parameters (rebuilt when they, the mode or the tap split change)."""
        from ..lgp import HipLGP
        # parameters AND BatchNorm buffers: version counters catch in-place edits (copy_, load_state_dict), data_ptr
        # catches re-assignment (.to(), .half(), p.data = ...) which does not bump _version
        key = (tuple(tap_channels), str(device), self.training,
               tuple((int(t._version), t.data_ptr()) for t in list(self.layers.parameters()) + list(self.layers.buffers())))
        if self._hip is None or self._hip_key != key:
            self._hip = HipLGP(self.state_dict(), tap_channels, device, training=self.training)
            self._hip_key = key
        return self._hip

    def _sync_running_stats(self):
        """Write the engine's BatchNorm running statistics back into the module (train-mode side effect the
        reference has because nobody calls .eval(): SURVEY Q3)."""
        if self._hip is None:
            return
        with torch.no_grad():
            for l, i in enumerate((2, 5, 8, 11)):
                bn = self.layers[i]
                bn.running_mean.copy_(self._hip.running_mean[l].to(bn.running_mean.dtype))
                bn.running_var.copy_(self._hip.running_var[l].to(bn.running_var.dtype))
                bn.num_batches_tracked.fill_(self._hip.num_batches_tracked[l])
        # the write-back above bumped the buffers' versions: re-key so the engine (which holds exactly these values)
        # is not rebuilt, while any LATER edit of parameters or buffers by the user still is
        if self._hip_key is not None:
            self._hip_key = self._hip_key[:3] + (
                tuple((int(t._version), t.data_ptr()) for t in list(self.layers.parameters()) + list(self.layers.buffers())),)

    def forward(self, x, t):
        """x (B, C, h, w) features, t (B, 4, h, w) noise level (the pipeline passes cat([nl] * 2)).
        Returns (B*w*h, out) fp16 in the reference's ``(b w h)`` row order.  B must be even ([uncond; cond]
        halves of B/2 samples: one BatchNorm batch per sample, as in B/2 separate reference calls) or 1."""
        from .. import ops
        assert x.is_cuda, "sketch2img_amd has no CPU path: move the inputs to the GPU"
        B, C, h, w = x.shape
        assert h == w, "square inputs only (SURVEY Q8)"
        in_dim = self.layers[0].in_features
        assert C + 4 + 4 * self.num_layers == in_dim and self.num_layers == 9
        if B % 2 == 1:
            assert B == 1
            x, t = torch.cat([x, x]), torch.cat([t, t])
        S = x.shape[0] // 2
        assert torch.equal(t[:S], t[S:]), "t must be cat([noise_level] * 2)"
        eng = self._engine([C], x.device)
        feats = x.float().permute(0, 2, 3, 1).reshape(-1, C).half().contiguous()
        out = eng.forward([(feats, h)], t[:S].float().contiguous(), 1.0, S, h)
        self._sync_running_stats()
        o = out[:, : self.layers[12].out_features].reshape(2 * S, h, w, -1).permute(0, 2, 1, 3)   # (b y x) -> (b w h)
        o = o.reshape(2 * S * w * h, -1)
        return o[: B * w * h].contiguous()


class FeatureTap:
    """What the reference's hook leaves on a UNet block: an object with ``.output`` (fp32 NCHW feature map of
    the last grad-enabled UNet evaluation; modules/latent_predictor.py:50-62)."""

    def __init__(self, name: str):
        self.name = name
        self._nhwc = None        # (fp16 [rows*s*s, C], rows, s)

    @property
    def output(self):
        if self._nhwc is None:
            raise AttributeError("output")          # the reference deletes it after use (pipeline.py:149)
        t, rows, s = self._nhwc
        return t.float().reshape(rows, s, s, -1).permute(0, 3, 1, 2).contiguous()

    @output.deleter
    def output(self):
        self._nhwc = None


TAP_NAMES = ["down_blocks.0", "down_blocks.1", "down_blocks.2", "mid_block.attentions.0", "mid_block.resnets.0",
             "mid_block.resnets.1", "up_blocks.0", "up_blocks.1", "up_blocks.2"]


def hook_unet(unet) -> List[FeatureTap]:
    """Same order as the reference (modules/latent_predictor.py:64-79).  ``unet`` is the HIP UNet facade
    (pipeline.unet); the taps are filled by every grad-enabled evaluation."""
    taps = [FeatureTap(n) for n in TAP_NAMES]
    unet._feature_taps = taps
    return taps

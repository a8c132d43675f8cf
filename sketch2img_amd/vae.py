"""Stable Diffusion VAE decoder on the libskg.so kernels.

Replaces what the reference reaches at modules/pipeline.py:118 (``decode_latents``: ``latents / 0.18215`` ->
``vae.decode(...).sample`` -> ``/2 + 0.5`` -> ``clamp(0, 1)`` -> NHWC fp32) with the AutoencoderKL that app.py:28-30
loads.  Architecture: oracle/vae.py (public SD-VAE config, validated by the decoder parameter count 49 490 179).

Same layout and kernels as the UNet: fp16 NHWC token-major activations [images*H*W, C]; 3x3 convolutions as implicit
GEMM (the nearest-2x upsample fused into the gather, CONV_UP2), GroupNorm(+SiLU) stats/apply, 1x1 shortcuts and the
attention projections as GEMMs.  The mid block's single-head attention (4096 tokens, head width 512 - outside the
flash kernel's head sizes, and 1.3 % of the decoder's FLOPs) runs per image as GEMM (q k^T, scale in the epilogue) ->
row softmax -> GEMM (p v).  Images are decoded in chunks so the 512x512x128-channel activations of the last level
(134 MB per image per tensor) stay bounded and inside the 2 GB buffer-descriptor range of the conv kernel.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from . import ops
from .config import VAEConfig, SD_VAE, vae_up_plan
from .unet import CIN_PAD, COUT_PAD, _h, _pad_vec, pack_conv


class HipVAEDecoder:
    def __init__(self, cfg: VAEConfig, state_dict: Dict[str, torch.Tensor], device="cuda", max_images_per_pass: int = 4):
        self.cfg, self.dev = cfg, torch.device(device)
        self.chunk = max_images_per_pass
        self.dtype = torch.float16
        self.W = self._pack(state_dict)

    # ------------------------------------------------------------------------------------------ weights
    def _pack(self, sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        cfg, dev = self.cfg, self.dev
        sd = {k: v for k, v in sd.items() if k.startswith(("decoder.", "post_quant_conv."))}
        W: Dict[str, torch.Tensor] = {}
        L = cfg.latent_channels
        for k, v in sd.items():
            v = v.detach().float()
            if k == "post_quant_conv.weight":
                # 1x1 conv on the CIN_PAD-wide latent rows: [CIN_PAD(out, zero rows beyond L), CIN_PAD(in)]
                w = torch.zeros(CIN_PAD, CIN_PAD)
                w[:L, :L] = v.reshape(L, L)
                W[k] = _h(w, dev)
            elif k == "post_quant_conv.bias":
                W[k] = _h(_pad_vec(v, CIN_PAD), dev)
            elif k == "decoder.conv_in.weight":
                W[k] = pack_conv(v, dev, cin_pad=CIN_PAD)
            elif k == "decoder.conv_out.weight":
                W[k] = pack_conv(v, dev, cout_pad=COUT_PAD)
            elif k == "decoder.conv_out.bias":
                W[k] = _h(_pad_vec(v, COUT_PAD), dev)
            elif v.ndim == 4 and v.shape[2] == 3:
                W[k] = pack_conv(v, dev)
            elif v.ndim == 4:                              # 1x1 conv_shortcut
                W[k] = _h(v.reshape(v.shape[0], v.shape[1]), dev)
            else:
                W[k] = _h(v, dev)
        a = "decoder.mid_block.attentions.0"
        W[a + ".qkv.weight"] = torch.cat([W.pop(f"{a}.{n}.weight") for n in ("query", "key", "value")]).contiguous()
        W[a + ".qkv.bias"] = torch.cat([W.pop(f"{a}.{n}.bias") for n in ("query", "key", "value")]).contiguous()
        return W

    def to(self, device):
        if torch.device(device) != self.dev:
            self.dev = torch.device(device)
            self.W = {k: v.to(self.dev) for k, v in self.W.items()}
        return self

    # ------------------------------------------------------------------------------------------ blocks
    def _res(self, p, x, rows, H):
        W, G, HW = self.W, self.cfg.norm_groups, H * H
        n1, _ = ops.groupnorm(x, rows, HW, G, 1e-6, W[p + ".norm1.weight"], W[p + ".norm1.bias"], True)
        h1 = ops.conv3x3(n1, W[p + ".conv1.weight"], rows, H, H, bias=W[p + ".conv1.bias"])
        del n1
        n2, _ = ops.groupnorm(h1, rows, HW, G, 1e-6, W[p + ".norm2.weight"], W[p + ".norm2.bias"], True)
        del h1
        if (p + ".conv_shortcut.weight") in W:
            x = ops.gemm(x, W[p + ".conv_shortcut.weight"], bias=W[p + ".conv_shortcut.bias"])
        return ops.conv3x3(n2, W[p + ".conv2.weight"], rows, H, H, bias=W[p + ".conv2.bias"], residual=x)

    def _attn(self, p, x, rows, H):
        W, G, HW = self.W, self.cfg.norm_groups, H * H
        C = x.shape[1]
        g, _ = ops.groupnorm(x, rows, HW, G, 1e-6, W[p + ".group_norm.weight"], W[p + ".group_norm.bias"], False)
        qkv = ops.gemm(g, W[p + ".qkv.weight"], bias=W[p + ".qkv.bias"])
        o = torch.empty(rows * HW, C, device=self.dev, dtype=torch.float16)
        scale = 1.0 / math.sqrt(C)
        for r in range(rows):
            blk = qkv[r * HW:(r + 1) * HW]
            s = ops.gemm(blk[:, :C], blk[:, C:2 * C], alpha=scale)          # [HW, HW] scores
            ops.softmax_rows(s, out=s)
            vt = ops.transpose(blk[:, 2 * C:])                               # [C, HW]
            ops.gemm(s, vt, out=o[r * HW:(r + 1) * HW])
        return ops.gemm(o, W[p + ".proj_attn.weight"], bias=W[p + ".proj_attn.bias"], residual=x)

    # ------------------------------------------------------------------------------------------ decode
    @torch.no_grad()
    def decode_tokens(self, z: torch.Tensor):
        """z float [S, 4, h, h] (already divided by the scaling factor) -> (fp16 NHWC [S*H*H, COUT_PAD], H)."""
        cfg, W = self.cfg, self.W
        S, _, h, _ = z.shape
        x = ops.nchw_to_nhwc(z.to(self.dev, torch.float32).contiguous(), CIN_PAD)
        x = ops.gemm(x, W["post_quant_conv.weight"], bias=W["post_quant_conv.bias"])
        x = ops.conv3x3(x, W["decoder.conv_in.weight"], S, h, h, bias=W["decoder.conv_in.bias"])
        x = self._res("decoder.mid_block.resnets.0", x, S, h)
        x = self._attn("decoder.mid_block.attentions.0", x, S, h)
        x = self._res("decoder.mid_block.resnets.1", x, S, h)
        H = h
        for i, res, up in vae_up_plan(cfg):
            for j in range(len(res)):
                x = self._res(f"decoder.up_blocks.{i}.resnets.{j}", x, S, H)
            if up:
                u = f"decoder.up_blocks.{i}.upsamplers.0.conv"
                x = ops.conv3x3(x, W[u + ".weight"], S, H, H, ops.CONV_UP2, bias=W[u + ".bias"])
                H *= 2
        n, _ = ops.groupnorm(x, S, H * H, cfg.norm_groups, 1e-6, W["decoder.conv_norm_out.weight"],
                             W["decoder.conv_norm_out.bias"], True)
        del x
        return ops.conv3x3(n, W["decoder.conv_out.weight"], S, H, H, bias=W["decoder.conv_out.bias"]), H

    @torch.no_grad()
    def decode(self, z: torch.Tensor) -> torch.Tensor:
        """AutoencoderKL.decode(z).sample: float NCHW [S, 3, 8h, 8h]."""
        outs = []
        for s0 in range(0, z.shape[0], self.chunk):
            y, H = self.decode_tokens(z[s0:s0 + self.chunk])
            outs.append(ops.nhwc_to_nchw(y, y.shape[0] // (H * H), self.cfg.out_channels, H, H))
        return torch.cat(outs)

    @torch.no_grad()
    def decode_latents(self, latents: torch.Tensor) -> torch.Tensor:
        """modules/pipeline.py:118: latents [S,4,h,h] -> float NHWC [S, 8h, 8h, 3] in [0, 1] (device tensor)."""
        outs = []
        for s0 in range(0, latents.shape[0], self.chunk):
            z = latents[s0:s0 + self.chunk].to(self.dev, torch.float32) * (1.0 / self.cfg.scaling_factor)
            y, H = self.decode_tokens(z)
            n = y.shape[0] // (H * H)
            outs.append(ops.image_postprocess(y, y.shape[0], self.cfg.out_channels).reshape(n, H, H, -1))
        return torch.cat(outs)


class _DecoderOutput:
    def __init__(self, sample):
        self.sample = sample


class AutoencoderKL:
    """Facade with the surface the reference uses of diffusers' AutoencoderKL (app.py:28-30,37: ``from_pretrained(path,
    subfolder="vae", torch_dtype=)``, passed as ``vae=`` to the pipeline; modules/pipeline.py:118 calls ``decode``
    through ``decode_latents``).  Decoder only: ``encode`` (app.py:109, the sketch target) is a next row."""

    def __init__(self, cfg: VAEConfig = SD_VAE, state_dict: Optional[Dict[str, torch.Tensor]] = None):
        from . import synthetic
        self.cfg = cfg
        self.config = cfg
        self._sd = state_dict if state_dict is not None else synthetic.vae_decoder_state_dict(cfg)
        self._hip: Optional[HipVAEDecoder] = None
        self.device = torch.device("cpu")
        self.dtype = torch.float16

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, subfolder: Optional[str] = None, torch_dtype=None,
                        config: Optional[VAEConfig] = None, **kwargs):
        import os
        sd = None
        if pretrained_model_name_or_path:
            d = os.path.join(pretrained_model_name_or_path, subfolder) if subfolder else pretrained_model_name_or_path
            st, pt = os.path.join(d, "diffusion_pytorch_model.safetensors"), os.path.join(d, "diffusion_pytorch_model.bin")
            if os.path.exists(st):
                from safetensors.torch import load_file
                sd = load_file(st)
            elif os.path.exists(pt):
                sd = torch.load(pt, map_location="cpu")
        return cls(config or SD_VAE, sd)

    def state_dict(self):
        return self._sd

    def to(self, device=None, dtype=None):
        if device is not None:
            self.device = torch.device(device)
            if self.device.type == "cuda":
                if self._hip is None:
                    self._hip = HipVAEDecoder(self.cfg, self._sd, self.device)
                else:
                    self._hip.to(self.device)
        return self

    @property
    def hip(self) -> HipVAEDecoder:
        if self._hip is None:
            raise RuntimeError("AutoencoderKL: call .to('cuda') first - the decoder runs on libskg.so kernels only")
        return self._hip

    def decode(self, z, return_dict: bool = True):
        y = self.hip.decode(z.float())
        return _DecoderOutput(y) if return_dict else (y,)

    def decode_latents(self, latents):
        return self.hip.decode_latents(latents)

    def encode(self, x):
        raise NotImplementedError("VAE encoder (app.py:109, sketch target) is not part of the hot path yet")

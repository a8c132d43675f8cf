"""Stable Diffusion VAE (decoder and encoder) on the libskg.so kernels.

Replaces what the reference reaches at modules/pipeline.py:118 (``decode_latents``: ``latents / 0.18215`` ->
``vae.decode(...).sample`` -> ``/2 + 0.5`` -> ``clamp(0, 1)`` -> NHWC fp32) with the AutoencoderKL that app.py:28-30
loads.  Architecture: oracle/vae.py (public SD-VAE config, validated by the decoder parameter count 49 490 179).

Same layout and kernels as the UNet: fp16 NHWC token-major activations [images*H*W, C]; 3x3 convolutions as implicit
GEMM (the nearest-2x upsample fused into the gather, CONV_UP2), GroupNorm(+SiLU) stats/apply, 1x1 shortcuts and the
attention projections as GEMMs.  The mid block's single-head attention (4096 tokens, head width 512 - outside the
flash kernel's head sizes, and 1.3 % of the decoder's FLOPs) runs per image as GEMM (q k^T, scale in the epilogue) ->
row softmax -> GEMM (p v).  Images are decoded in chunks so the 512x512x128-channel activations of the last level
(134 MB per image per tensor) stay bounded and inside the 2 GB buffer-descriptor range of the conv kernel.

The encoder (app.py:109, ``vae.encode(img).latent_dist.sample() * 0.18215`` = the sketch target) is the mirror image:
conv_in, 4 levels of 2 resnets with a stride-2 convolution padded (0,1,0,1) in between (CONV_S2A gather), the same
mid block, GroupNorm+SiLU, conv_out with quant_conv (1x1) folded into its weights, and a small kernel that turns the
moments into a sample.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch

from . import ops
from .config import VAEConfig, SD_VAE, vae_up_plan
from .unet import CIN_PAD, COUT_PAD, UP2_POLYPHASE, _h, _pad_vec, pack_conv, pack_conv_up2


class _HipVAEBlocks:
    """Resnet / attention blocks shared by the decoder and the encoder (weights in self.W, config in self.cfg)."""

    def to(self, device):
        if torch.device(device) != self.dev:
            self.dev = torch.device(device)
            self.W = {k: v.to(self.dev) for k, v in self.W.items()}
        return self

    def _fuse_qkv(self, W, a):
        W[a + ".qkv.weight"] = torch.cat([W.pop(f"{a}.{n}.weight") for n in ("query", "key", "value")]).contiguous()
        W[a + ".qkv.bias"] = torch.cat([W.pop(f"{a}.{n}.bias") for n in ("query", "key", "value")]).contiguous()

    @staticmethod
    def normalise_attention_keys(sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        """The VAE mid-block attention under either diffusers spelling: legacy AttentionBlock
        (``query / key / value / proj_attn``, the 0.12-0.14 era the reference dates to) or the current Attention class
        (``to_q / to_k / to_v / to_out.0``, e.g. sd-vae-ft-mse re-saves; Linear or 1x1-conv shaped) -> legacy names,
        2-D weights."""
        ren = {"to_q": "query", "to_k": "key", "to_v": "value", "to_out.0": "proj_attn"}
        out = {}
        for k, v in sd.items():
            if ".attentions." in k:
                for new, old in ren.items():
                    if f".{new}." in k:
                        k = k.replace(f".{new}.", f".{old}.")
                        break
                if k.endswith(".weight") and v.dim() == 4 and any(f".{o}." in k for o in ren.values()):
                    v = v.reshape(v.shape[0], v.shape[1])
            out[k] = v
        return out

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



class HipVAEDecoder(_HipVAEBlocks):
    def __init__(self, cfg: VAEConfig, state_dict: Dict[str, torch.Tensor], device="cuda", max_images_per_pass: int = 4):
        self.cfg, self.dev = cfg, torch.device(device)
        self.chunk = max_images_per_pass
        self.dtype = torch.float16
        self.W = self._pack(state_dict)

    # ------------------------------------------------------------------------------------------ weights
    def _pack(self, sd: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        sd = self.normalise_attention_keys(sd)
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
                if ".upsamplers." in k and UP2_POLYPHASE and v.shape[1] % 64 == 0:
                    W[k + ":pp"] = pack_conv_up2(v, dev)
            elif v.ndim == 4:                              # 1x1 conv_shortcut
                W[k] = _h(v.reshape(v.shape[0], v.shape[1]), dev)
            else:
                W[k] = _h(v, dev)
        self._fuse_qkv(W, "decoder.mid_block.attentions.0")
        return W

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
                if (u + ".weight:pp") in W:      # polyphase: 16 instead of 36 tap-products per low-res pixel
                    x = ops.conv_up2(x, W[u + ".weight:pp"], S, H, H, bias=W[u + ".bias"], W9=W.get(u + ".weight"))
                else:
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


    @torch.no_grad()
    def decode_to_u8(self, latents: torch.Tensor) -> torch.Tensor:
        """decode_latents + numpy_to_pil's quantisation on the device: latents [S,4,h,h] -> uint8 [S, 8h, 8h, 3]
        (what a rank hands to the final gather: 786 432 B per 512x512 image)."""
        outs = []
        for s0 in range(0, latents.shape[0], self.chunk):
            z = latents[s0:s0 + self.chunk].to(self.dev, torch.float32) * (1.0 / self.cfg.scaling_factor)
            y, H = self.decode_tokens(z)
            n = y.shape[0] // (H * H)
            outs.append(ops.image_to_u8(y, y.shape[0], self.cfg.out_channels).reshape(n, H, H, -1))
        return torch.cat(outs)


class HipVAEEncoder(_HipVAEBlocks):
    """AutoencoderKL.encode on the HIP kernels: float images [S, 3, H, W] in [-1, 1] -> fp16 NHWC moments."""

    def __init__(self, cfg: VAEConfig, state_dict: Dict[str, torch.Tensor], device="cuda", max_images_per_pass: int = 4):
        self.cfg, self.dev = cfg, torch.device(device)
        self.chunk = max_images_per_pass
        self.W = self._pack(state_dict)

    def _pack(self, sd):
        sd = self.normalise_attention_keys(sd)
        cfg, dev = self.cfg, self.dev
        sd = {k: v.detach().float() for k, v in sd.items() if k.startswith(("encoder.", "quant_conv."))}
        W: Dict[str, torch.Tensor] = {}
        L2 = 2 * cfg.latent_channels
        # quant_conv (1x1, after conv_out, nothing in between) folds exactly into conv_out: W' = Wq Wout, b' = Wq b + bq
        wq = sd.pop("quant_conv.weight").reshape(L2, L2)
        bq = sd.pop("quant_conv.bias")
        wo, bo = sd.pop("encoder.conv_out.weight"), sd.pop("encoder.conv_out.bias")
        W["encoder.conv_out.weight"] = pack_conv(torch.einsum("qo,oikl->qikl", wq, wo), dev)
        W["encoder.conv_out.bias"] = _h(wq @ bo + bq, dev)
        for k, v in sd.items():
            if k == "encoder.conv_in.weight":
                W[k] = pack_conv(v, dev, cin_pad=CIN_PAD)
            elif v.ndim == 4 and v.shape[2] == 3:
                W[k] = pack_conv(v, dev)
            elif v.ndim == 4:
                W[k] = _h(v.reshape(v.shape[0], v.shape[1]), dev)
            else:
                W[k] = _h(v, dev)
        self._fuse_qkv(W, "encoder.mid_block.attentions.0")
        return W

    @torch.no_grad()
    def moments(self, img: torch.Tensor):
        """img float [S, 3, H, W] -> (fp16 NHWC moments [S*h*h, 2*latent], h) with h = H / 8 (4-level layout)."""
        cfg, W = self.cfg, self.W
        S, _, H, Wd = img.shape
        if H != Wd or H % 2 ** (len(cfg.block_out_channels) - 1):
            raise ValueError(f"VAE encoder: square images with a side divisible by 8, got {tuple(img.shape)}")
        x = ops.nchw_to_nhwc(img.to(self.dev, torch.float32).contiguous(), CIN_PAD)
        x = ops.conv3x3(x, W["encoder.conv_in.weight"], S, H, H, bias=W["encoder.conv_in.bias"])
        nb = len(cfg.block_out_channels)
        for i in range(nb):
            for j in range(cfg.layers_per_block):
                x = self._res(f"encoder.down_blocks.{i}.resnets.{j}", x, S, H)
            if i != nb - 1:
                d = f"encoder.down_blocks.{i}.downsamplers.0.conv"
                x = ops.conv3x3(x, W[d + ".weight"], S, H, H, ops.CONV_S2A, bias=W[d + ".bias"])
                H //= 2
        x = self._res("encoder.mid_block.resnets.0", x, S, H)
        x = self._attn("encoder.mid_block.attentions.0", x, S, H)
        x = self._res("encoder.mid_block.resnets.1", x, S, H)
        n, _ = ops.groupnorm(x, S, H * H, cfg.norm_groups, 1e-6, W["encoder.conv_norm_out.weight"],
                             W["encoder.conv_norm_out.bias"], True)
        return ops.conv3x3(n, W["encoder.conv_out.weight"], S, H, H, bias=W["encoder.conv_out.bias"]), H

    @torch.no_grad()
    def encode(self, img: torch.Tensor, noise: Optional[torch.Tensor] = None, scale: float = 1.0) -> torch.Tensor:
        """latent_dist.sample() * scale with the N(0,1) draw `noise` [S,4,h,h] (None: latent_dist.mode() * scale)."""
        outs = []
        L = self.cfg.latent_channels
        for s0 in range(0, img.shape[0], self.chunk):
            m, h = self.moments(img[s0:s0 + self.chunk])
            n = m.shape[0] // (h * h)
            nz = None if noise is None else noise[s0:s0 + n].to(self.dev, torch.float32).contiguous()
            outs.append(ops.gaussian_sample(m, n, L, h * h, nz, scale).reshape(n, L, h, h))
        return torch.cat(outs)


class _LatentDist:
    """DiagonalGaussianDistribution surface used by app.py:109 (.sample()), plus .mode() / .mean / .logvar / .std."""

    def __init__(self, enc: HipVAEEncoder, img: torch.Tensor):
        self._enc, self._img = enc, img
        self._mom = None

    def _moments(self):
        if self._mom is None:
            ms, L = [], self._enc.cfg.latent_channels
            for s0 in range(0, self._img.shape[0], self._enc.chunk):
                m, h = self._enc.moments(self._img[s0:s0 + self._enc.chunk])
                ms.append(ops.nhwc_to_nchw(m, m.shape[0] // (h * h), 2 * L, h, h))
            self._mom = torch.cat(ms)
        return self._mom

    @property
    def mean(self):
        return self._moments()[:, :self._enc.cfg.latent_channels]

    @property
    def logvar(self):
        return self._moments()[:, self._enc.cfg.latent_channels:].clamp(-30.0, 20.0)

    @property
    def std(self):
        return torch.exp(0.5 * self.logvar)

    def mode(self):
        return self._enc.encode(self._img)

    def sample(self, generator: Optional[torch.Generator] = None):
        S, _, H, _ = self._img.shape
        h = H // 2 ** (len(self._enc.cfg.block_out_channels) - 1)
        dev = generator.device if generator is not None else self._enc.dev
        noise = torch.randn(S, self._enc.cfg.latent_channels, h, h, generator=generator, device=dev)
        return self._enc.encode(self._img, noise)


class _EncoderOutput:
    def __init__(self, latent_dist):
        self.latent_dist = latent_dist


class _DecoderOutput:
    def __init__(self, sample):
        self.sample = sample


class AutoencoderKL:
    """Facade with the surface the reference uses of diffusers' AutoencoderKL (app.py:28-30,37: ``from_pretrained(path,
    subfolder="vae", torch_dtype=)``, passed as ``vae=`` to the pipeline; modules/pipeline.py:118 calls ``decode``
    through ``decode_latents``; app.py:109 calls ``encode(img).latent_dist.sample()`` for the sketch target)."""

    def __init__(self, cfg: VAEConfig = SD_VAE, state_dict: Optional[Dict[str, torch.Tensor]] = None):
        from . import synthetic
        self.cfg = cfg
        self.config = cfg
        if state_dict is None:
            state_dict = dict(synthetic.vae_decoder_state_dict(cfg))
            state_dict.update(synthetic.vae_encoder_state_dict(cfg))
        self._sd = state_dict
        self._hip: Optional[HipVAEDecoder] = None
        self._hip_enc: Optional[HipVAEEncoder] = None
        self.device = torch.device("cpu")
        self.dtype = torch.float16

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path=None, subfolder: Optional[str] = None, torch_dtype=None,
                        config: Optional[VAEConfig] = None, **kwargs):
        """``None``: seeded synthetic weights (explicit opt-in).  Any other path must resolve to a local diffusers-layout
        folder with the weights, else FileNotFoundError - never a silent fall-through to random weights."""
        import os
        sd = None
        if pretrained_model_name_or_path is not None:
            from .modules.pipeline import load_diffusers_weights
            d = os.path.join(pretrained_model_name_or_path, subfolder) if subfolder else pretrained_model_name_or_path
            sd = load_diffusers_weights(d, "AutoencoderKL.from_pretrained")
        return cls(config or SD_VAE, sd)

    def state_dict(self):
        return self._sd

    def to(self, device=None, dtype=None):
        if device is not None:
            self.device = torch.device(device)
            if self.device.type == "cuda":
                if self._hip is None:
                    self._hip = HipVAEDecoder(self.cfg, self._sd, self.device)
                    if any(k.startswith("encoder.") for k in self._sd):
                        self._hip_enc = HipVAEEncoder(self.cfg, self._sd, self.device)
                else:
                    self._hip.to(self.device)
                    if self._hip_enc is not None:
                        self._hip_enc.to(self.device)
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

    def decode_to_u8(self, latents):
        return self.hip.decode_to_u8(latents)

    def encode(self, x, return_dict: bool = True):
        self.hip
        if self._hip_enc is None:
            raise RuntimeError("AutoencoderKL.encode: the state_dict has no encoder.* weights")
        d = _LatentDist(self._hip_enc, x.float())
        return _EncoderOutput(d) if return_dict else (d,)

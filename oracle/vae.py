"""Oracle: Stable Diffusion VAE (AutoencoderKL.decode / .encode), plain PyTorch, NCHW.  TEST INFRASTRUCTURE.

Restates what the reference reaches at ``modules/pipeline.py:118`` (``self.decode_latents(latents)``: third-party
StableDiffusionPipeline.decode_latents = ``latents / 0.18215`` -> ``vae.decode(...).sample`` -> ``/2 + 0.5`` ->
``clamp(0, 1)`` -> NHWC fp32 numpy) with the VAE that ``app.py:28-30`` loads
(``AutoencoderKL.from_pretrained("runwayml/stable-diffusion-v1-5", subfolder="vae")``).  The arithmetic lives in
third-party ``diffusers`` (AutoencoderKL / Decoder / UNetMidBlock2D / AttentionBlock / UpDecoderBlock2D /
ResnetBlock2D / Upsample2D, 0.12.x-0.14.x by API usage), absent here: PARITY UNPINNED.  The architecture followed is
the public SD-VAE config (block_out_channels 128/256/512/512, layers_per_block 2, latent_channels 4, norm groups 32,
GroupNorm eps 1e-6, single-head mid attention) and is validated by the exact decoder parameter count
49 490 179 (+ 20 for post_quant_conv; the full AutoencoderKL has 83 653 863).

Decoder:  post_quant_conv 1x1 4->4;  conv_in 3x3 4->C3;  mid: Res(C3), Attn(C3), Res(C3);
          up blocks over reversed(block_out): (layers_per_block + 1) x Res, then nearest-2x + conv3x3 except the last;
          GroupNorm -> SiLU -> conv_out 3x3 C0->3.
Res(ci->co): GN -> SiLU -> conv3x3 -> GN -> SiLU -> conv3x3, + (1x1 conv_shortcut if ci != co), no time embedding.
Attn(C):  GN -> q,k,v Linear(C,C)+b over the HW tokens, ONE head of width C: softmax(q k^T / sqrt(C)) v ->
          proj_attn Linear(C,C)+b -> + residual.  Scores / probabilities are kept in the activation dtype with an
          fp32 softmax, as AttentionBlock does.

Encoder (``app.py:109``: ``vae.encode(img).latent_dist.sample() * 0.18215`` makes the sketch target; third-party
diffusers Encoder / DownEncoderBlock2D / Downsample2D / DiagonalGaussianDistribution, PARITY UNPINNED; validated by
the exact parameter count 34 163 592 + 72 for quant_conv):
          conv_in 3x3 3->C0;  down blocks over block_out: layers_per_block x Res, then (except the last)
          F.pad(0,1,0,1) + conv3x3 stride 2 padding 0;  mid: Res, Attn, Res;  GroupNorm -> SiLU ->
          conv_out 3x3 C3->2*latent;  quant_conv 1x1;  moments = (mean, logvar): logvar clamped to [-30, 20],
          sample = mean + exp(0.5 logvar) * noise, mode = mean.

Weights are a flat dict with the diffusers AutoencoderKL state_dict key names (``decoder.*``, ``post_quant_conv.*``,
``encoder.*``, ``quant_conv.*``).
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Dict, Tuple

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class VAEConfig:
    latent_channels: int = 4
    out_channels: int = 3
    block_out_channels: Tuple[int, ...] = (128, 256, 512, 512)
    layers_per_block: int = 2
    norm_groups: int = 32
    scaling_factor: float = 0.18215


SD_VAE = VAEConfig()
TINY_VAE = VAEConfig(block_out_channels=(32, 64, 64, 64), layers_per_block=1, norm_groups=8)


def _res_shapes(p: str, ci: int, co: int, out):
    out[p + ".norm1.weight"] = (ci,)
    out[p + ".norm1.bias"] = (ci,)
    out[p + ".conv1.weight"] = (co, ci, 3, 3)
    out[p + ".conv1.bias"] = (co,)
    out[p + ".norm2.weight"] = (co,)
    out[p + ".norm2.bias"] = (co,)
    out[p + ".conv2.weight"] = (co, co, 3, 3)
    out[p + ".conv2.bias"] = (co,)
    if ci != co:
        out[p + ".conv_shortcut.weight"] = (co, ci, 1, 1)
        out[p + ".conv_shortcut.bias"] = (co,)


def up_plan(cfg: VAEConfig):
    """[(block index, [(cin, cout) per resnet], has_upsampler)] in execution order."""
    rev = list(reversed(cfg.block_out_channels))
    plan, prev = [], rev[0]
    for i, co in enumerate(rev):
        res = []
        for j in range(cfg.layers_per_block + 1):
            res.append((prev if j == 0 else co, co))
        plan.append((i, res, i != len(rev) - 1))
        prev = co
    return plan


def decoder_param_shapes(cfg: VAEConfig = SD_VAE) -> "OrderedDict[str, tuple]":
    out: "OrderedDict[str, tuple]" = OrderedDict()
    c_top = cfg.block_out_channels[-1]
    out["post_quant_conv.weight"] = (cfg.latent_channels, cfg.latent_channels, 1, 1)
    out["post_quant_conv.bias"] = (cfg.latent_channels,)
    out["decoder.conv_in.weight"] = (c_top, cfg.latent_channels, 3, 3)
    out["decoder.conv_in.bias"] = (c_top,)
    _res_shapes("decoder.mid_block.resnets.0", c_top, c_top, out)
    a = "decoder.mid_block.attentions.0"
    out[a + ".group_norm.weight"] = (c_top,)
    out[a + ".group_norm.bias"] = (c_top,)
    for n in ("query", "key", "value", "proj_attn"):
        out[f"{a}.{n}.weight"] = (c_top, c_top)
        out[f"{a}.{n}.bias"] = (c_top,)
    _res_shapes("decoder.mid_block.resnets.1", c_top, c_top, out)
    for i, res, up in up_plan(cfg):
        for j, (ci, co) in enumerate(res):
            _res_shapes(f"decoder.up_blocks.{i}.resnets.{j}", ci, co, out)
        if up:
            co = res[-1][1]
            out[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (co, co, 3, 3)
            out[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (co,)
    c0 = cfg.block_out_channels[0]
    out["decoder.conv_norm_out.weight"] = (c0,)
    out["decoder.conv_norm_out.bias"] = (c0,)
    out["decoder.conv_out.weight"] = (cfg.out_channels, c0, 3, 3)
    out["decoder.conv_out.bias"] = (cfg.out_channels,)
    return out


def init_weights(cfg: VAEConfig = SD_VAE, seed: int = 20260930) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights: U(+-1/sqrt(fan_in)) for conv / linear, gamma 1 + small noise, small biases
    (fp16-representable values so the HIP path and the oracle start from identical numbers)."""
    g = torch.Generator().manual_seed(seed)
    W: Dict[str, torch.Tensor] = {}
    for k, shp in decoder_param_shapes(cfg).items():
        if "norm" in k and k.endswith(".weight"):
            w = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            w = 0.05 * torch.randn(shp, generator=g)
        else:
            fan_in = math.prod(shp[1:])
            w = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(fan_in)
        W[k] = w.half().float()
    return W


def _gn(W, p, x, groups):
    return F.group_norm(x, groups, W[p + ".weight"], W[p + ".bias"], eps=1e-6)


def _resnet(W, p, x, groups):
    h = F.conv2d(F.silu(_gn(W, p + ".norm1", x, groups)), W[p + ".conv1.weight"], W[p + ".conv1.bias"], padding=1)
    h = F.conv2d(F.silu(_gn(W, p + ".norm2", h, groups)), W[p + ".conv2.weight"], W[p + ".conv2.bias"], padding=1)
    if (p + ".conv_shortcut.weight") in W:
        x = F.conv2d(x, W[p + ".conv_shortcut.weight"], W[p + ".conv_shortcut.bias"])
    return x + h


def _attn(W, p, x, groups):
    B, C, H, Wd = x.shape
    t = _gn(W, p + ".group_norm", x, groups).reshape(B, C, H * Wd).transpose(1, 2)       # (B, HW, C)
    q = F.linear(t, W[p + ".query.weight"], W[p + ".query.bias"])
    k = F.linear(t, W[p + ".key.weight"], W[p + ".key.bias"])
    v = F.linear(t, W[p + ".value.weight"], W[p + ".value.bias"])
    s = torch.bmm(q, k.transpose(1, 2)) * (1.0 / math.sqrt(C))
    a = torch.softmax(s.float(), dim=-1).to(s.dtype)
    o = F.linear(torch.bmm(a, v), W[p + ".proj_attn.weight"], W[p + ".proj_attn.bias"])
    return x + o.transpose(1, 2).reshape(B, C, H, Wd)


def decode(cfg: VAEConfig, W: Dict[str, torch.Tensor], z: torch.Tensor) -> torch.Tensor:
    """AutoencoderKL.decode(z).sample: z (B, 4, h, w) -> (B, 3, 8h, 8w) for the 4-level SD layout."""
    g = cfg.norm_groups
    x = F.conv2d(z, W["post_quant_conv.weight"], W["post_quant_conv.bias"])
    x = F.conv2d(x, W["decoder.conv_in.weight"], W["decoder.conv_in.bias"], padding=1)
    x = _resnet(W, "decoder.mid_block.resnets.0", x, g)
    x = _attn(W, "decoder.mid_block.attentions.0", x, g)
    x = _resnet(W, "decoder.mid_block.resnets.1", x, g)
    for i, res, up in up_plan(cfg):
        for j in range(len(res)):
            x = _resnet(W, f"decoder.up_blocks.{i}.resnets.{j}", x, g)
        if up:
            x = F.interpolate(x, scale_factor=2.0, mode="nearest")
            x = F.conv2d(x, W[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"],
                         W[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"], padding=1)
    x = F.silu(_gn(W, "decoder.conv_norm_out", x, g))
    return F.conv2d(x, W["decoder.conv_out.weight"], W["decoder.conv_out.bias"], padding=1)


def decode_latents(cfg: VAEConfig, W: Dict[str, torch.Tensor], latents: torch.Tensor) -> torch.Tensor:
    """modules/pipeline.py:118 -> (B, H, W, 3) fp32 in [0, 1] (the numpy array the pipeline turns into PIL)."""
    img = decode(cfg, W, (1.0 / cfg.scaling_factor) * latents)
    return (img / 2 + 0.5).clamp(0, 1).permute(0, 2, 3, 1).float()


def decoder_flops(cfg: VAEConfig, h: int) -> float:
    """Multiply-add x 2 of one decode of an h x h latent (convs, attention, linears)."""
    fl = 0.0
    shapes = decoder_param_shapes(cfg)
    c_top = cfg.block_out_channels[-1]
    res_at = {"decoder.conv_in": h, "decoder.mid_block": h, "post_quant_conv": h}
    cur = h
    for i, res, up in up_plan(cfg):
        res_at[f"decoder.up_blocks.{i}.resnets"] = cur
        if up:
            cur *= 2
            res_at[f"decoder.up_blocks.{i}.upsamplers"] = cur
    res_at["decoder.conv_out"] = cur
    for k, shp in shapes.items():
        if not k.endswith(".weight") or len(shp) < 2:
            continue
        side = next(v for p, v in res_at.items() if k.startswith(p))
        fl += 2.0 * math.prod(shp) * side * side
    fl += 2.0 * 2.0 * (h * h) ** 2 * c_top          # q k^T and p v
    return fl


# ------------------------------------------------------------------------------------------------------ encoder
def encoder_param_shapes(cfg: VAEConfig = SD_VAE) -> "OrderedDict[str, tuple]":
    out: "OrderedDict[str, tuple]" = OrderedDict()
    boc, L = cfg.block_out_channels, cfg.latent_channels
    out["encoder.conv_in.weight"] = (boc[0], cfg.out_channels, 3, 3)
    out["encoder.conv_in.bias"] = (boc[0],)
    prev = boc[0]
    for i, co in enumerate(boc):
        for j in range(cfg.layers_per_block):
            _res_shapes(f"encoder.down_blocks.{i}.resnets.{j}", prev if j == 0 else co, co, out)
        if i != len(boc) - 1:
            out[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (co, co, 3, 3)
            out[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (co,)
        prev = co
    ct = boc[-1]
    _res_shapes("encoder.mid_block.resnets.0", ct, ct, out)
    a = "encoder.mid_block.attentions.0"
    out[a + ".group_norm.weight"] = (ct,)
    out[a + ".group_norm.bias"] = (ct,)
    for n in ("query", "key", "value", "proj_attn"):
        out[f"{a}.{n}.weight"] = (ct, ct)
        out[f"{a}.{n}.bias"] = (ct,)
    _res_shapes("encoder.mid_block.resnets.1", ct, ct, out)
    out["encoder.conv_norm_out.weight"] = (ct,)
    out["encoder.conv_norm_out.bias"] = (ct,)
    out["encoder.conv_out.weight"] = (2 * L, ct, 3, 3)
    out["encoder.conv_out.bias"] = (2 * L,)
    out["quant_conv.weight"] = (2 * L, 2 * L, 1, 1)
    out["quant_conv.bias"] = (2 * L,)
    return out


def init_encoder_weights(cfg: VAEConfig = SD_VAE, seed: int = 20261001) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    W: Dict[str, torch.Tensor] = {}
    for k, shp in encoder_param_shapes(cfg).items():
        if "norm" in k and k.endswith(".weight"):
            w = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            w = 0.05 * torch.randn(shp, generator=g)
        else:
            w = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(math.prod(shp[1:]))
        W[k] = w.half().float()
    return W


def encode_moments(cfg: VAEConfig, W: Dict[str, torch.Tensor], img: torch.Tensor):
    """AutoencoderKL.encode(img).latent_dist -> (mean, logvar clamped), each (B, 4, H/8, W/8)."""
    g = cfg.norm_groups
    x = F.conv2d(img, W["encoder.conv_in.weight"], W["encoder.conv_in.bias"], padding=1)
    nb = len(cfg.block_out_channels)
    for i in range(nb):
        for j in range(cfg.layers_per_block):
            x = _resnet(W, f"encoder.down_blocks.{i}.resnets.{j}", x, g)
        if i != nb - 1:
            x = F.conv2d(F.pad(x, (0, 1, 0, 1)), W[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"],
                         W[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"], stride=2)
    x = _resnet(W, "encoder.mid_block.resnets.0", x, g)
    x = _attn(W, "encoder.mid_block.attentions.0", x, g)
    x = _resnet(W, "encoder.mid_block.resnets.1", x, g)
    x = F.silu(_gn(W, "encoder.conv_norm_out", x, g))
    x = F.conv2d(x, W["encoder.conv_out.weight"], W["encoder.conv_out.bias"], padding=1)
    m = F.conv2d(x, W["quant_conv.weight"], W["quant_conv.bias"])
    mean, logvar = m.chunk(2, dim=1)
    return mean, logvar.clamp(-30.0, 20.0)


def encode_sample(cfg: VAEConfig, W, img: torch.Tensor, noise: torch.Tensor) -> torch.Tensor:
    """latent_dist.sample() with the standard-normal draw passed in (the reference draws it from torch's RNG)."""
    mean, logvar = encode_moments(cfg, W, img)
    return mean + torch.exp(0.5 * logvar) * noise

"""Seeded synthetic weights and inputs (there are no checkpoints or datasets on the build / GPU boxes).

Recipe: SURVEY.md section 8(d).  UNet / LGP tensors in diffusers / reference state_dict key order from one
``torch.Generator``; Linear / conv U(+-1/sqrt(fan_in)); LGP kaiming-uniform with zero bias as
modules/latent_predictor.py:32-35; text embeddings, initial latents and sketch targets from fixed seeds.
All values are rounded through fp16 so that an fp32 checker sees bit-identical parameters.
"""
from __future__ import annotations

import hashlib
import math
import os
from collections import OrderedDict
from typing import Dict, Tuple

import torch

from .config import CLIPVisionConfig, UNetConfig, VAEConfig, tap_channels, up_block_plan, vae_up_plan

WEIGHT_SEED = 20260929


def _resnet(p, cin, cout, temb, s):
    s[p + ".norm1.weight"] = (cin,); s[p + ".norm1.bias"] = (cin,)
    s[p + ".conv1.weight"] = (cout, cin, 3, 3); s[p + ".conv1.bias"] = (cout,)
    s[p + ".time_emb_proj.weight"] = (cout, temb); s[p + ".time_emb_proj.bias"] = (cout,)
    s[p + ".norm2.weight"] = (cout,); s[p + ".norm2.bias"] = (cout,)
    s[p + ".conv2.weight"] = (cout, cout, 3, 3); s[p + ".conv2.bias"] = (cout,)
    if cin != cout:
        s[p + ".conv_shortcut.weight"] = (cout, cin, 1, 1); s[p + ".conv_shortcut.bias"] = (cout,)


def _attn(p, c, ctx, linear, s):
    s[p + ".norm.weight"] = (c,); s[p + ".norm.bias"] = (c,)
    pw = (c, c) if linear else (c, c, 1, 1)
    s[p + ".proj_in.weight"] = pw; s[p + ".proj_in.bias"] = (c,)
    t = p + ".transformer_blocks.0"
    for n in ("norm1", "norm2", "norm3"):
        s[f"{t}.{n}.weight"] = (c,); s[f"{t}.{n}.bias"] = (c,)
    for a, kd in (("attn1", c), ("attn2", ctx)):
        s[f"{t}.{a}.to_q.weight"] = (c, c); s[f"{t}.{a}.to_k.weight"] = (c, kd); s[f"{t}.{a}.to_v.weight"] = (c, kd)
        s[f"{t}.{a}.to_out.0.weight"] = (c, c); s[f"{t}.{a}.to_out.0.bias"] = (c,)
    s[f"{t}.ff.net.0.proj.weight"] = (8 * c, c); s[f"{t}.ff.net.0.proj.bias"] = (8 * c,)
    s[f"{t}.ff.net.2.weight"] = (c, 4 * c); s[f"{t}.ff.net.2.bias"] = (c,)
    s[p + ".proj_out.weight"] = pw; s[p + ".proj_out.bias"] = (c,)


def unet_param_shapes(cfg: UNetConfig) -> "OrderedDict[str, tuple]":
    """diffusers UNet2DConditionModel state_dict keys -> shapes for an SD-layout config."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    boc, temb, nb = cfg.block_out_channels, cfg.time_embed_dim, len(cfg.block_out_channels)
    s["conv_in.weight"] = (boc[0], cfg.in_channels, 3, 3); s["conv_in.bias"] = (boc[0],)
    s["time_embedding.linear_1.weight"] = (temb, boc[0]); s["time_embedding.linear_1.bias"] = (temb,)
    s["time_embedding.linear_2.weight"] = (temb, temb); s["time_embedding.linear_2.bias"] = (temb,)
    cin = boc[0]
    for i, cout in enumerate(boc):
        for j in range(cfg.layers_per_block):
            _resnet(f"down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, temb, s)
            if i < nb - 1:
                _attn(f"down_blocks.{i}.attentions.{j}", cout, cfg.cross_attention_dim, cfg.use_linear_projection, s)
        if i < nb - 1:
            s[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            s[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (cout,)
        cin = cout
    for i, blk in enumerate(up_block_plan(cfg)):
        for j, (hin, skip, cout) in enumerate(blk):
            _resnet(f"up_blocks.{i}.resnets.{j}", hin + skip, cout, temb, s)
            if i > 0:
                _attn(f"up_blocks.{i}.attentions.{j}", cout, cfg.cross_attention_dim, cfg.use_linear_projection, s)
        if i < nb - 1:
            c = blk[0][2]
            s[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3)
            s[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
    cm = boc[-1]
    _resnet("mid_block.resnets.0", cm, cm, temb, s)
    _attn("mid_block.attentions.0", cm, cfg.cross_attention_dim, cfg.use_linear_projection, s)
    _resnet("mid_block.resnets.1", cm, cm, temb, s)
    s["conv_norm_out.weight"] = (boc[0],); s["conv_norm_out.bias"] = (boc[0],)
    s["conv_out.weight"] = (cfg.out_channels, boc[0], 3, 3); s["conv_out.bias"] = (cfg.out_channels,)
    return s


def _disk_cached(name: str, make):
    """SKG_SYNTH_CACHE_DIR=<dir> (the test suite sets it: a dozen bench.py subprocesses each draw the same 860 M values, 10-13 s a
    time): the tensors are stored as fp16 - every value is fp16-representable by construction, so the round trip is exact."""
    d = os.environ.get("SKG_SYNTH_CACHE_DIR")
    if not d:
        return make()
    path = os.path.join(d, name + ".pt")
    if os.path.exists(path):
        return {k: v.float() for k, v in torch.load(path, mmap=True).items()}
    sd = make()
    tmp = f"{path}.{os.getpid()}.tmp"
    torch.save({k: v.half() for k, v in sd.items()}, tmp)
    os.replace(tmp, path)
    return sd


def unet_state_dict(cfg: UNetConfig, seed: int = WEIGHT_SEED) -> Dict[str, torch.Tensor]:
    shapes = unet_param_shapes(cfg)
    if sum(math.prod(s) for s in shapes.values()) > 100_000_000:      # (full-size architectures only: the tiny test configs take milliseconds)
        tag = hashlib.sha256(repr((cfg, seed, sorted(shapes.items()))).encode()).hexdigest()[:16]
        return _disk_cached(f"unet_{tag}", lambda: _unet_state_dict(cfg, seed, shapes))
    return _unet_state_dict(cfg, seed, shapes)


def _unet_state_dict(cfg: UNetConfig, seed: int, shapes) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, torch.Tensor] = {}
    for k, shp in shapes.items():
        leaf = k.split(".")[-2]
        if leaf.startswith("norm") or leaf == "conv_norm_out":
            w = (torch.ones(shp) if k.endswith("weight") else torch.zeros(shp)) + 0.1 * (torch.rand(shp, generator=g) - 0.5)
        else:
            wk = k[: -len("bias")] + "weight" if k.endswith("bias") else k
            bound = 1.0 / math.sqrt(math.prod(shapes[wk][1:]))
            w = (torch.rand(shp, generator=g) * 2 - 1) * bound
        sd[k] = w.half().float()
    return sd


LGP_LIN, LGP_BN, LGP_HIDDEN = (0, 3, 6, 9, 12), (2, 5, 8, 11), (512, 256, 128, 64)


def lgp_state_dict(input_dim: int = 9320, output_dim: int = 4, seed: int = WEIGHT_SEED,
                   perturb_bn: bool = True) -> Dict[str, torch.Tensor]:
    """The reference's LatentEdgePredictor(input_dim, output_dim, 9).state_dict() layout (30 keys)."""
    g = torch.Generator().manual_seed(seed)
    dims = (input_dim,) + LGP_HIDDEN + (output_dim,)
    sd: Dict[str, torch.Tensor] = {}
    for i in range(5):
        bound = math.sqrt(6.0 / dims[i])
        sd[f"layers.{LGP_LIN[i]}.weight"] = ((torch.rand(dims[i + 1], dims[i], generator=g) * 2 - 1) * bound).half().float()
        sd[f"layers.{LGP_LIN[i]}.bias"] = torch.zeros(dims[i + 1])
        if i < 4:
            b, n = LGP_BN[i], dims[i + 1]
            pert = lambda: 0.2 * (torch.rand(n, generator=g) - 0.5) if perturb_bn else torch.zeros(n)
            sd[f"layers.{b}.weight"] = (torch.ones(n) + pert()).half().float()
            sd[f"layers.{b}.bias"] = pert().half().float()
            sd[f"layers.{b}.running_mean"] = torch.zeros(n)
            sd[f"layers.{b}.running_var"] = torch.ones(n)
            sd[f"layers.{b}.num_batches_tracked"] = torch.zeros((), dtype=torch.int64)
    return sd


def text_embeddings(samples: int, dim: int = 768, tokens: int = 77) -> torch.Tensor:
    """[uncond rows (seed 8); cond rows (seed 7)], every sample uses the same fixed 'prompt'."""
    gu, gc = torch.Generator().manual_seed(8), torch.Generator().manual_seed(7)
    u = torch.randn(1, tokens, dim, generator=gu).half().float()
    c = torch.randn(1, tokens, dim, generator=gc).half().float()
    return torch.cat([u.expand(samples, -1, -1), c.expand(samples, -1, -1)]).contiguous()


def initial_latents(first: int, count: int, h: int) -> torch.Tensor:
    """Per-sample CPU generators, seed 1000 + global sample index (placement independent)."""
    return torch.cat([torch.randn(1, 4, h, h, generator=torch.Generator().manual_seed(1000 + i))
                      for i in range(first, first + count)])


def sketch_targets(first: int, count: int, h: int) -> torch.Tensor:
    """Stand-in for vae.encode(sketch) * 0.18215 (app.py:109): 8 seeded random polylines on an 8h x 8h
    canvas, average-pooled 8x, replicated to 4 channels, mapped to [-1, 1] * 0.18215."""
    out = []
    for i in range(first, first + count):
        g = torch.Generator().manual_seed(2000 + i)
        n = 8 * h
        img = torch.zeros(n, n)
        for _ in range(8):
            pts = (torch.rand(4, 2, generator=g) * (n - 1))
            for a, b in zip(pts[:-1], pts[1:]):
                steps = int(max(abs(float(b[0] - a[0])), abs(float(b[1] - a[1])))) + 1
                tt = torch.linspace(0, 1, steps)
                ys = (a[0] + (b[0] - a[0]) * tt).round().long().clamp(0, n - 1)
                xs = (a[1] + (b[1] - a[1]) * tt).round().long().clamp(0, n - 1)
                for dy in (-1, 0, 1):
                    for dx in (-1, 0, 1):
                        img[(ys + dy).clamp(0, n - 1), (xs + dx).clamp(0, n - 1)] = 1.0
        pooled = torch.nn.functional.avg_pool2d(img[None, None], 8)[0, 0]
        out.append(((1.0 - pooled) * 2 - 1).expand(4, -1, -1) * 0.18215)     # white paper, dark strokes
    return torch.stack(out).contiguous()


def lgp_input_dim(cfg: UNetConfig) -> int:
    return sum(tap_channels(cfg)) + 4 + 36


# ------------------------------------------------------------------------- injected attention (configs 4 / 5)
SAT_WEIGHT_SEED = 20260930


def satmixin_param_shapes(cfg: UNetConfig, variant: str) -> "OrderedDict[str, tuple]":
    """SatMixin.state_dict() keys / shapes (modules/clip_guided_attn.py:14-27,52-63; sketch_guided_attn.py:52-70):
    per BasicTransformerBlock, in unet.named_modules() order; variant 'clip' adds sketch_proj."""
    from .inject import block_dims, module_name
    m: "OrderedDict[str, tuple]" = OrderedDict()
    for path, c, _ in block_dims(cfg):
        n = module_name(path)
        if variant == "clip":
            m[f"{n}.sketch_proj.weight"] = (c, 1024)
            m[f"{n}.sketch_proj.bias"] = (c,)
        m[f"{n}.sketch_norm.weight"] = (c,)
        m[f"{n}.sketch_norm.bias"] = (c,)
        for q in ("to_q", "to_k", "to_v"):
            m[f"{n}.sketch_attn.{q}.weight"] = (c, c)
        m[f"{n}.sketch_attn.to_out.0.weight"] = (c, c)
        m[f"{n}.sketch_attn.to_out.0.bias"] = (c,)
        m[f"{n}.sketch_conv.weight"] = (c, c, 1)
        m[f"{n}.sketch_conv.bias"] = (c,)
    return m


def satmixin_state_dict(cfg: UNetConfig, variant: str, seed: int = SAT_WEIGHT_SEED) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    man = satmixin_param_shapes(cfg, variant)
    sd: Dict[str, torch.Tensor] = {}
    for k, shp in man.items():
        if ".sketch_norm." in k:
            v = (torch.ones(shp) if k.endswith("weight") else torch.zeros(shp)) + 0.1 * (torch.rand(shp, generator=g) - 0.5)
        else:
            wk = k[: -len("bias")] + "weight" if k.endswith("bias") else k
            v = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(math.prod(man[wk][1:]))
        sd[k] = v.half().float()
    return sd


def res_samples(cfg: UNetConfig, first: int, count: int, h: int):
    """Config 4 input: what ``SatMixin.set_res_samples`` takes (modules/sketch_guided_attn.py:29-40) - per
    down block a tuple of NCHW tensors, here seeded per sample (seed 3000 + global sample index) and laid out in the
    sampler's row order [uncond rows ; cond rows] with both CFG rows of a sample carrying the same features (the
    SketchEncoder sees the same sketch for both)."""
    boc, nb = cfg.block_out_channels, len(cfg.block_out_channels)
    gens = [torch.Generator().manual_seed(3000 + i) for i in range(first, first + count)]
    out = []
    for i, c in enumerate(boc):
        sizes = [h >> i] * cfg.layers_per_block + ([h >> (i + 1)] if i < nb - 1 else [])
        blk = []
        for s in sizes:
            one = torch.stack([torch.randn(c, s, s, generator=g) for g in gens]).half().float()
            blk.append(torch.cat([one, one]))
        out.append(tuple(blk))
    return out


def sketch_state(first: int, count: int, tokens: int = 257, dim: int = 1024) -> torch.Tensor:
    """Config 5 input: [zeros (uncond rows) ; CLIP-vision hidden states (cond rows)] as modules/clip_guided_inf.py:107
    stacks them; the hidden states are seeded stand-ins, seed 9 + global sample index (seed 9)."""
    hid = torch.stack([torch.randn(tokens, dim, generator=torch.Generator().manual_seed(9 + i))
                       for i in range(first, first + count)]).half().float()
    return torch.cat([torch.zeros_like(hid), hid])


# ---------------------------------------------------------------------------------------------- VAE decoder
VAE_WEIGHT_SEED = 20260930


def _vae_res(p, ci, co, s):
    s[p + ".norm1.weight"] = (ci,); s[p + ".norm1.bias"] = (ci,)
    s[p + ".conv1.weight"] = (co, ci, 3, 3); s[p + ".conv1.bias"] = (co,)
    s[p + ".norm2.weight"] = (co,); s[p + ".norm2.bias"] = (co,)
    s[p + ".conv2.weight"] = (co, co, 3, 3); s[p + ".conv2.bias"] = (co,)
    if ci != co:
        s[p + ".conv_shortcut.weight"] = (co, ci, 1, 1); s[p + ".conv_shortcut.bias"] = (co,)


def vae_decoder_param_shapes(cfg: VAEConfig) -> "OrderedDict[str, tuple]":
    """diffusers AutoencoderKL state_dict keys (decoder.*, post_quant_conv.*) -> shapes."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    ct, L = cfg.block_out_channels[-1], cfg.latent_channels
    s["post_quant_conv.weight"] = (L, L, 1, 1); s["post_quant_conv.bias"] = (L,)
    s["decoder.conv_in.weight"] = (ct, L, 3, 3); s["decoder.conv_in.bias"] = (ct,)
    _vae_res("decoder.mid_block.resnets.0", ct, ct, s)
    a = "decoder.mid_block.attentions.0"
    s[a + ".group_norm.weight"] = (ct,); s[a + ".group_norm.bias"] = (ct,)
    for n in ("query", "key", "value", "proj_attn"):
        s[f"{a}.{n}.weight"] = (ct, ct); s[f"{a}.{n}.bias"] = (ct,)
    _vae_res("decoder.mid_block.resnets.1", ct, ct, s)
    for i, res, up in vae_up_plan(cfg):
        for j, (ci, co) in enumerate(res):
            _vae_res(f"decoder.up_blocks.{i}.resnets.{j}", ci, co, s)
        if up:
            co = res[-1][1]
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.weight"] = (co, co, 3, 3)
            s[f"decoder.up_blocks.{i}.upsamplers.0.conv.bias"] = (co,)
    c0 = cfg.block_out_channels[0]
    s["decoder.conv_norm_out.weight"] = (c0,); s["decoder.conv_norm_out.bias"] = (c0,)
    s["decoder.conv_out.weight"] = (cfg.out_channels, c0, 3, 3); s["decoder.conv_out.bias"] = (cfg.out_channels,)
    return s


def vae_decoder_state_dict(cfg: VAEConfig, seed: int = VAE_WEIGHT_SEED) -> Dict[str, torch.Tensor]:
    """Seeded synthetic decoder weights (same recipe and draw order as oracle/vae.py init_weights, which the
    parity tests use as the checker's copy): conv / linear U(+-1/sqrt(fan_in)), gamma 1 + 0.1 N, biases 0.05 N."""
    g = torch.Generator().manual_seed(seed)
    W: Dict[str, torch.Tensor] = {}
    for k, shp in vae_decoder_param_shapes(cfg).items():
        if "norm" in k and k.endswith(".weight"):
            w = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            w = 0.05 * torch.randn(shp, generator=g)
        else:
            w = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(math.prod(shp[1:]))
        W[k] = w.half().float()
    return W


VAE_ENC_WEIGHT_SEED = 20261001


def vae_encoder_param_shapes(cfg: VAEConfig) -> "OrderedDict[str, tuple]":
    """diffusers AutoencoderKL state_dict keys (encoder.*, quant_conv.*) -> shapes."""
    s: "OrderedDict[str, tuple]" = OrderedDict()
    boc, L = cfg.block_out_channels, cfg.latent_channels
    s["encoder.conv_in.weight"] = (boc[0], cfg.out_channels, 3, 3); s["encoder.conv_in.bias"] = (boc[0],)
    prev = boc[0]
    for i, co in enumerate(boc):
        for j in range(cfg.layers_per_block):
            _vae_res(f"encoder.down_blocks.{i}.resnets.{j}", prev if j == 0 else co, co, s)
        if i != len(boc) - 1:
            s[f"encoder.down_blocks.{i}.downsamplers.0.conv.weight"] = (co, co, 3, 3)
            s[f"encoder.down_blocks.{i}.downsamplers.0.conv.bias"] = (co,)
        prev = co
    ct = boc[-1]
    _vae_res("encoder.mid_block.resnets.0", ct, ct, s)
    a = "encoder.mid_block.attentions.0"
    s[a + ".group_norm.weight"] = (ct,); s[a + ".group_norm.bias"] = (ct,)
    for n in ("query", "key", "value", "proj_attn"):
        s[f"{a}.{n}.weight"] = (ct, ct); s[f"{a}.{n}.bias"] = (ct,)
    _vae_res("encoder.mid_block.resnets.1", ct, ct, s)
    s["encoder.conv_norm_out.weight"] = (ct,); s["encoder.conv_norm_out.bias"] = (ct,)
    s["encoder.conv_out.weight"] = (2 * L, ct, 3, 3); s["encoder.conv_out.bias"] = (2 * L,)
    s["quant_conv.weight"] = (2 * L, 2 * L, 1, 1); s["quant_conv.bias"] = (2 * L,)
    return s


def vae_encoder_state_dict(cfg: VAEConfig, seed: int = VAE_ENC_WEIGHT_SEED) -> Dict[str, torch.Tensor]:
    g = torch.Generator().manual_seed(seed)
    W: Dict[str, torch.Tensor] = {}
    for k, shp in vae_encoder_param_shapes(cfg).items():
        if "norm" in k and k.endswith(".weight"):
            w = 1.0 + 0.1 * torch.randn(shp, generator=g)
        elif k.endswith(".bias"):
            w = 0.05 * torch.randn(shp, generator=g)
        else:
            w = (torch.rand(shp, generator=g) * 2 - 1) / math.sqrt(math.prod(shp[1:]))
        W[k] = w.half().float()
    return W


# ---------------------------------------------------------------------------------------------- CLIP vision tower
CLIP_WEIGHT_SEED = 20261002


def clip_vision_param_shapes(cfg: CLIPVisionConfig) -> "OrderedDict[str, tuple]":
    """transformers CLIPVisionModel state_dict keys (without the 4.x ``vision_model.`` prefix) -> shapes."""
    D, I = cfg.hidden_size, cfg.intermediate_size
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["embeddings.class_embedding"] = (D,)
    s["embeddings.patch_embedding.weight"] = (D, 3, cfg.patch_size, cfg.patch_size)
    s["embeddings.position_embedding.weight"] = (cfg.num_tokens, D)
    s["pre_layrnorm.weight"] = (D,); s["pre_layrnorm.bias"] = (D,)
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{l}"
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[f"{p}.self_attn.{n}.weight"] = (D, D); s[f"{p}.self_attn.{n}.bias"] = (D,)
        s[f"{p}.layer_norm1.weight"] = (D,); s[f"{p}.layer_norm1.bias"] = (D,)
        s[f"{p}.mlp.fc1.weight"] = (I, D); s[f"{p}.mlp.fc1.bias"] = (I,)
        s[f"{p}.mlp.fc2.weight"] = (D, I); s[f"{p}.mlp.fc2.bias"] = (D,)
        s[f"{p}.layer_norm2.weight"] = (D,); s[f"{p}.layer_norm2.bias"] = (D,)
    s["post_layernorm.weight"] = (D,); s["post_layernorm.bias"] = (D,)
    return s


def clip_vision_state_dict(cfg: CLIPVisionConfig, seed: int = CLIP_WEIGHT_SEED) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights, same recipe and draw order as oracle/clip_vision.py init_weights."""
    g = torch.Generator().manual_seed(seed)
    W: Dict[str, torch.Tensor] = {}
    for k, shp in clip_vision_param_shapes(cfg).items():
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


# ---------------------------------------------------------------------------------------------- CLIP text encoder
CLIP_TEXT_WEIGHT_SEED = 20261003


def clip_text_param_shapes(cfg) -> "OrderedDict[str, tuple]":
    """transformers CLIPTextModel state_dict keys (without the 4.x ``text_model.`` prefix) -> shapes."""
    D, I = cfg.hidden_size, cfg.intermediate_size
    s: "OrderedDict[str, tuple]" = OrderedDict()
    s["embeddings.token_embedding.weight"] = (cfg.vocab_size, D)
    s["embeddings.position_embedding.weight"] = (cfg.max_position_embeddings, D)
    for l in range(cfg.num_hidden_layers):
        p = f"encoder.layers.{l}"
        for n in ("k_proj", "v_proj", "q_proj", "out_proj"):
            s[f"{p}.self_attn.{n}.weight"] = (D, D); s[f"{p}.self_attn.{n}.bias"] = (D,)
        s[f"{p}.layer_norm1.weight"] = (D,); s[f"{p}.layer_norm1.bias"] = (D,)
        s[f"{p}.mlp.fc1.weight"] = (I, D); s[f"{p}.mlp.fc1.bias"] = (I,)
        s[f"{p}.mlp.fc2.weight"] = (D, I); s[f"{p}.mlp.fc2.bias"] = (D,)
        s[f"{p}.layer_norm2.weight"] = (D,); s[f"{p}.layer_norm2.bias"] = (D,)
    s["final_layer_norm.weight"] = (D,); s["final_layer_norm.bias"] = (D,)
    return s


def clip_text_state_dict(cfg, seed: int = CLIP_TEXT_WEIGHT_SEED) -> Dict[str, torch.Tensor]:
    """Seeded synthetic weights, same recipe and draw order as oracle/clip_text.py init_weights."""
    g = torch.Generator().manual_seed(seed)
    W: Dict[str, torch.Tensor] = {}
    for k, shp in clip_text_param_shapes(cfg).items():
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

"""Oracle: SD-style conditional UNet (epsilon network), plain PyTorch, NCHW.  TEST INFRASTRUCTURE.

Restates what the reference reaches at ``modules/pipeline.py:96``
(``self.unet(latent_model_input, t, encoder_hidden_states=...)``) and taps at
``modules/latent_predictor.py:47-81``.  The arithmetic lives in third-party ``diffusers``
(UNet2DConditionModel, ResnetBlock2D, Transformer2DModel, BasicTransformerBlock,
CrossAttention; 0.12.x-0.14.x by API usage) which is absent here: PARITY UNPINNED, see
``oracle/__init__.py``.  The architecture description followed is SURVEY.md section 8 (a3);
it is validated by the exact parameter counts 859 520 964 (SD1.5) / 865 910 724 (SD2.1).

Weights are a flat ``dict[str, Tensor]`` using the diffusers state_dict key names, so a
diffusers-layout checkpoint loads unchanged.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F


@dataclass(frozen=True)
class UNetConfig:
    in_channels: int = 4
    out_channels: int = 4
    block_out_channels: Tuple[int, ...] = (320, 640, 1280, 1280)
    layers_per_block: int = 2
    cross_attention_dim: int = 768
    # SD1.5: 8 heads everywhere (diffusers' "attention_head_dim": 8 is a head COUNT).
    # SD2.1: (5, 10, 20, 20) heads -> head_dim 64.
    num_heads: Tuple[int, ...] = (8, 8, 8, 8)
    use_linear_projection: bool = False
    norm_groups: int = 32
    sample_size: int = 64
    # blocks 0..n-2 are CrossAttnDown (with downsampler), the last is a plain DownBlock
    # (SD layout); mirrored for the up path.

    @property
    def time_embed_dim(self) -> int:
        return self.block_out_channels[0] * 4


SD15 = UNetConfig()
SD21 = UNetConfig(cross_attention_dim=1024, num_heads=(5, 10, 20, 20), use_linear_projection=True,
                  sample_size=96)
# small config used by fast parity tests (same topology, narrow channels)
TINY = UNetConfig(block_out_channels=(32, 64, 128, 128), cross_attention_dim=64,
                  num_heads=(2, 2, 4, 4), norm_groups=8, sample_size=32)


# --------------------------------------------------------------------------------------
# parameter inventory
# --------------------------------------------------------------------------------------
def _resnet_shapes(p: str, cin: int, cout: int, temb: int, out: "OrderedDict[str, tuple]"):
    out[p + ".norm1.weight"] = (cin,)
    out[p + ".norm1.bias"] = (cin,)
    out[p + ".conv1.weight"] = (cout, cin, 3, 3)
    out[p + ".conv1.bias"] = (cout,)
    out[p + ".time_emb_proj.weight"] = (cout, temb)
    out[p + ".time_emb_proj.bias"] = (cout,)
    out[p + ".norm2.weight"] = (cout,)
    out[p + ".norm2.bias"] = (cout,)
    out[p + ".conv2.weight"] = (cout, cout, 3, 3)
    out[p + ".conv2.bias"] = (cout,)
    if cin != cout:
        out[p + ".conv_shortcut.weight"] = (cout, cin, 1, 1)
        out[p + ".conv_shortcut.bias"] = (cout,)


def _attn_shapes(p: str, c: int, ctx: int, linear: bool, out: "OrderedDict[str, tuple]"):
    out[p + ".norm.weight"] = (c,)
    out[p + ".norm.bias"] = (c,)
    pw = (c, c) if linear else (c, c, 1, 1)
    out[p + ".proj_in.weight"] = pw
    out[p + ".proj_in.bias"] = (c,)
    t = p + ".transformer_blocks.0"
    for n in ("norm1", "norm2", "norm3"):
        out[f"{t}.{n}.weight"] = (c,)
        out[f"{t}.{n}.bias"] = (c,)
    for a, kd in (("attn1", c), ("attn2", ctx)):
        out[f"{t}.{a}.to_q.weight"] = (c, c)
        out[f"{t}.{a}.to_k.weight"] = (c, kd)
        out[f"{t}.{a}.to_v.weight"] = (c, kd)
        out[f"{t}.{a}.to_out.0.weight"] = (c, c)
        out[f"{t}.{a}.to_out.0.bias"] = (c,)
    out[f"{t}.ff.net.0.proj.weight"] = (8 * c, c)
    out[f"{t}.ff.net.0.proj.bias"] = (8 * c,)
    out[f"{t}.ff.net.2.weight"] = (c, 4 * c)
    out[f"{t}.ff.net.2.bias"] = (c,)
    out[p + ".proj_out.weight"] = pw
    out[p + ".proj_out.bias"] = (c,)


def up_block_plan(cfg: UNetConfig) -> List[List[Tuple[int, int, int]]]:
    """Per up block, per resnet: (h_channels_in, skip_channels, out_channels)."""
    rev = tuple(reversed(cfg.block_out_channels))
    nb = len(rev)
    plan = []
    for i in range(nb):
        out_c = rev[i]
        prev = rev[i - 1] if i > 0 else rev[0]
        inp = rev[min(i + 1, nb - 1)]
        blk = []
        for j in range(cfg.layers_per_block + 1):
            skip = inp if j == cfg.layers_per_block else out_c
            hin = prev if j == 0 else out_c
            blk.append((hin, skip, out_c))
        plan.append(blk)
    return plan


def param_shapes(cfg: UNetConfig) -> "OrderedDict[str, tuple]":
    s: "OrderedDict[str, tuple]" = OrderedDict()
    boc = cfg.block_out_channels
    temb = cfg.time_embed_dim
    nb = len(boc)
    s["conv_in.weight"] = (boc[0], cfg.in_channels, 3, 3)
    s["conv_in.bias"] = (boc[0],)
    s["time_embedding.linear_1.weight"] = (temb, boc[0])
    s["time_embedding.linear_1.bias"] = (temb,)
    s["time_embedding.linear_2.weight"] = (temb, temb)
    s["time_embedding.linear_2.bias"] = (temb,)
    cin = boc[0]
    for i, cout in enumerate(boc):
        has_attn = i < nb - 1
        for j in range(cfg.layers_per_block):
            _resnet_shapes(f"down_blocks.{i}.resnets.{j}", cin if j == 0 else cout, cout, temb, s)
            if has_attn:
                _attn_shapes(f"down_blocks.{i}.attentions.{j}", cout, cfg.cross_attention_dim,
                             cfg.use_linear_projection, s)
        if i < nb - 1:
            s[f"down_blocks.{i}.downsamplers.0.conv.weight"] = (cout, cout, 3, 3)
            s[f"down_blocks.{i}.downsamplers.0.conv.bias"] = (cout,)
        cin = cout
    plan = up_block_plan(cfg)
    for i, blk in enumerate(plan):
        has_attn = i > 0
        for j, (hin, skip, cout) in enumerate(blk):
            _resnet_shapes(f"up_blocks.{i}.resnets.{j}", hin + skip, cout, temb, s)
            if has_attn:
                _attn_shapes(f"up_blocks.{i}.attentions.{j}", cout, cfg.cross_attention_dim,
                             cfg.use_linear_projection, s)
        if i < nb - 1:
            c = blk[0][2]
            s[f"up_blocks.{i}.upsamplers.0.conv.weight"] = (c, c, 3, 3)
            s[f"up_blocks.{i}.upsamplers.0.conv.bias"] = (c,)
    cm = boc[-1]
    _resnet_shapes("mid_block.resnets.0", cm, cm, temb, s)
    _attn_shapes("mid_block.attentions.0", cm, cfg.cross_attention_dim, cfg.use_linear_projection, s)
    _resnet_shapes("mid_block.resnets.1", cm, cm, temb, s)
    s["conv_norm_out.weight"] = (boc[0],)
    s["conv_norm_out.bias"] = (boc[0],)
    s["conv_out.weight"] = (cfg.out_channels, boc[0], 3, 3)
    s["conv_out.bias"] = (cfg.out_channels,)
    return s


def param_count(cfg: UNetConfig) -> int:
    return sum(math.prod(v) for v in param_shapes(cfg).values())


def init_weights(cfg: UNetConfig, seed: int = 20260929, dtype=torch.float32,
                 round_fp16: bool = True) -> Dict[str, torch.Tensor]:
    """Synthetic seeded weights: Linear/conv U(+-1/sqrt(fan_in)), norm
    gamma=1 beta=0, biases U(+-1/sqrt(fan_in)).  Values are rounded through fp16 so the fp16
    HIP path and the fp32 oracle hold bit-identical parameters."""
    g = torch.Generator().manual_seed(seed)
    W: Dict[str, torch.Tensor] = {}
    shapes = param_shapes(cfg)
    for k, shp in shapes.items():
        leaf = k.split(".")[-2]
        if leaf.startswith("norm") or leaf == "conv_norm_out":
            w = torch.ones(shp) if k.endswith("weight") else torch.zeros(shp)
            # perturb norms a little so gamma/beta handling is actually exercised
            w = w + 0.1 * (torch.rand(shp, generator=g) - 0.5)
        else:
            wk = k[: -len("bias")] + "weight" if k.endswith("bias") else k
            fan_in = math.prod(shapes[wk][1:])
            bound = 1.0 / math.sqrt(fan_in)
            w = (torch.rand(shp, generator=g) * 2 - 1) * bound
        if round_fp16:
            w = w.half().float()
        W[k] = w.to(dtype)
    return W


# --------------------------------------------------------------------------------------
# forward
# --------------------------------------------------------------------------------------
# fp16-storage emulation.  The reference runs its UNet in fp16 on the GPU (app.py:34 torch_dtype=float16):
# every tensor an fp16 pipeline writes to memory is rounded to fp16 while the arithmetic in between
# (convolution / matmul accumulation, normalisation statistics, softmax) stays in fp32.  Inside
# ``with fp16_storage():`` the oracle rounds at exactly those tensor boundaries (``_r``), which splits
# the distance between the HIP path and this fp32 oracle into
#     (HIP  vs  fp16-storage oracle)   = kernel error proper (accumulation order, exp2 / gelu forms,
#                                         fp16 P operand of the PV product, pre-scaled Q), and
#     (fp16-storage oracle  vs  fp32)  = what ANY fp16 implementation, the reference's included, pays.
# Autograd sees ``x.half().float()``: its backward rounds the gradient at the same boundaries.
# Every rounding point carries a kind so that tools/eps_decompose.py can switch classes of tensors off one at a
# time ("which stored tensors cost the accuracy"): "res" = the residual stream (ResnetBlock / attention / FF sums,
# conv_in / resampling outputs), "norm" = GroupNorm(+SiLU) / LayerNorm outputs, "lin" = conv / Linear outputs that
# feed a norm or an activation, "attn" = q / k / v and attention outputs, "temb" = the time-embedding path, "rop" = the residual stream where it is
# itself a matmul OPERAND (rop_sc: conv_shortcut, rop_dn / rop_up: the resampling convolutions, rop_po: proj_out; skip=("rop",) covers all; idempotent when
# "res" is rounded anyway).
_FP16_STORAGE = False
_FP16_SKIP: frozenset = frozenset()


class fp16_storage:
    def __init__(self, on: bool = True, skip=()):
        self.on, self.skip = on, frozenset(skip)

    def __enter__(self):
        global _FP16_STORAGE, _FP16_SKIP
        self.prev = (_FP16_STORAGE, _FP16_SKIP)
        _FP16_STORAGE, _FP16_SKIP = self.on, self.skip
        return self

    def __exit__(self, *exc):
        global _FP16_STORAGE, _FP16_SKIP
        _FP16_STORAGE, _FP16_SKIP = self.prev


def _r(x: torch.Tensor, kind: str = "lin") -> torch.Tensor:
    """kind "lin_n" / "lin_a": a conv / Linear output consumed by a norm or a sum (lin_n: conv1 -> norm2, conv_shortcut ->
    residual sum) or by an activation / matmul (lin_a: FF1 -> GEGLU -> FF2); skip=("lin",) covers both."""
    if not _FP16_STORAGE or kind in _FP16_SKIP or kind.split("_")[0] in _FP16_SKIP:
        return x
    return x.half().to(x.dtype)


def timestep_embedding(t: torch.Tensor, dim: int) -> torch.Tensor:
    """diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=0), fp32."""
    half = dim // 2
    exponent = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
    emb = t.float()[:, None] * torch.exp(exponent)[None, :]
    return torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)


def _gn(x, W, p, groups, eps):
    return F.group_norm(x, groups, W[p + ".weight"], W[p + ".bias"], eps)


def resnet_forward(cfg, W, p, x, temb_act):
    h = _r(F.silu(_gn(x, W, p + ".norm1", cfg.norm_groups, 1e-5)), "norm")
    tproj = _r(F.linear(temb_act, W[p + ".time_emb_proj.weight"], W[p + ".time_emb_proj.bias"]), "temb")
    tb = _r(tproj + W[p + ".conv1.bias"], "temb")
    h = _r(F.conv2d(h, W[p + ".conv1.weight"], None, padding=1) + tb[:, :, None, None], "lin_n")
    h = _r(F.silu(_gn(h, W, p + ".norm2", cfg.norm_groups, 1e-5)), "norm")
    h = F.conv2d(h, W[p + ".conv2.weight"], W[p + ".conv2.bias"], padding=1)
    if (p + ".conv_shortcut.weight") in W:
        x = _r(F.conv2d(_r(x, "rop_sc"), W[p + ".conv_shortcut.weight"], W[p + ".conv_shortcut.bias"]), "lin_n")
    return _r(x + h, "res")


def attention(q, k, v, heads):
    """softmax(q k^T / sqrt(d)) v with ``heads`` heads; q (B,N,C), k/v (B,M,C)."""
    B, N, C = q.shape
    d = C // heads
    qh = q.view(B, N, heads, d).transpose(1, 2)
    kh = k.view(B, -1, heads, d).transpose(1, 2)
    vh = v.view(B, -1, heads, d).transpose(1, 2)
    if B * heads * N * kh.shape[2] > (1 << 28):
        # config 5 (9216 + 257 tokens): one (row, head) pair at a time keeps the score matrix at 0.36 GB
        o = torch.stack([torch.stack([torch.matmul(torch.softmax(torch.matmul(qh[b, i], kh[b, i].t()) * (d ** -0.5), dim=-1),
                                                   vh[b, i]) for i in range(heads)]) for b in range(B)])
    else:
        s = torch.matmul(qh, kh.transpose(-1, -2)) * (d ** -0.5)
        o = torch.matmul(torch.softmax(s, dim=-1), vh)
    return _r(o.transpose(1, 2).reshape(B, N, C), "attn")


def cross_attention_module(W, p, x, ctx, heads):
    """diffusers CrossAttention: to_q/to_k/to_v without bias, to_out.0 with bias."""
    ctx = x if ctx is None else ctx
    q = _r(F.linear(x, W[p + ".to_q.weight"]), "attn")
    k = _r(F.linear(ctx, W[p + ".to_k.weight"]), "attn")
    v = _r(F.linear(ctx, W[p + ".to_v.weight"]), "attn")
    o = attention(q, k, v, heads)
    return F.linear(o, W[p + ".to_out.0.weight"], W[p + ".to_out.0.bias"])      # caller adds the residual, then rounds


InjectFn = Callable[[str, torch.Tensor, int], torch.Tensor]


def transformer_block_forward(W, p, x, ehs, heads, inject: Optional[InjectFn] = None):
    """BasicTransformerBlock.  ``inject(path, hidden_states, heads)`` is the reference's
    step 1.5 (modules/clip_guided_attn.py:111-125, modules/sketch_guided_attn.py:120-132),
    applied between self- and cross-attention; it returns the new hidden states."""
    c = x.shape[-1]
    n = _r(F.layer_norm(x, (c,), W[p + ".norm1.weight"], W[p + ".norm1.bias"], 1e-5), "norm")
    x = _r(cross_attention_module(W, p + ".attn1", n, None, heads) + x, "res")
    if inject is not None:
        x = inject(p, x, heads)
    n = _r(F.layer_norm(x, (c,), W[p + ".norm2.weight"], W[p + ".norm2.bias"], 1e-5), "norm")
    x = _r(cross_attention_module(W, p + ".attn2", n, ehs, heads) + x, "res")
    n = _r(F.layer_norm(x, (c,), W[p + ".norm3.weight"], W[p + ".norm3.bias"], 1e-5), "norm")
    hcat = _r(F.linear(n, W[p + ".ff.net.0.proj.weight"], W[p + ".ff.net.0.proj.bias"]), "lin_a")
    hid, gate = hcat.chunk(2, dim=-1)
    ff = F.linear(_r(hid * F.gelu(gate), "lin_a"), W[p + ".ff.net.2.weight"], W[p + ".ff.net.2.bias"])
    return _r(ff + x, "res")


def transformer2d_forward(cfg, W, p, x, ehs, heads, inject=None):
    B, C, H, Wd = x.shape
    res = x
    h = _r(_gn(x, W, p + ".norm", cfg.norm_groups, 1e-6), "norm")
    if cfg.use_linear_projection:
        h = h.permute(0, 2, 3, 1).reshape(B, H * Wd, C)
        h = _r(F.linear(h, W[p + ".proj_in.weight"], W[p + ".proj_in.bias"]), "res")
    else:
        h = _r(F.conv2d(h, W[p + ".proj_in.weight"], W[p + ".proj_in.bias"]), "res")
        h = h.permute(0, 2, 3, 1).reshape(B, H * Wd, C)
    h = _r(transformer_block_forward(W, p + ".transformer_blocks.0", h, ehs, heads, inject), "rop_po")
    if cfg.use_linear_projection:
        h = F.linear(h, W[p + ".proj_out.weight"], W[p + ".proj_out.bias"])
        h = h.reshape(B, H, Wd, C).permute(0, 3, 1, 2)
    else:
        h = h.reshape(B, H, Wd, C).permute(0, 3, 1, 2)
        h = F.conv2d(h, W[p + ".proj_out.weight"], W[p + ".proj_out.bias"])
    return _r(h + res, "res")


def transformer_block_paths(cfg: UNetConfig) -> List[str]:
    """BasicTransformerBlock paths in ``unet.named_modules()`` order: down, up, mid
    (module registration order of UNet2DConditionModel)."""
    nb = len(cfg.block_out_channels)
    out = []
    for i in range(nb - 1):
        for j in range(cfg.layers_per_block):
            out.append(f"down_blocks.{i}.attentions.{j}.transformer_blocks.0")
    for i in range(1, nb):
        for j in range(cfg.layers_per_block + 1):
            out.append(f"up_blocks.{i}.attentions.{j}.transformer_blocks.0")
    out.append("mid_block.attentions.0.transformer_blocks.0")
    return out


def unet_forward(cfg: UNetConfig, W: Dict[str, torch.Tensor], x: torch.Tensor, t,
                 ehs: torch.Tensor, inject: Optional[InjectFn] = None,
                 down_only: bool = False):
    """Returns (eps, taps).  taps = the 9 feature maps the reference hooks, in its concat order
    (modules/latent_predictor.py:64-79): down_blocks[0..2] outputs (after the downsampler),
    mid_block.attentions[0], mid_block.resnets[0], mid_block.resnets[1], up_blocks[0..2]
    outputs (after the upsampler).  ``down_only`` mirrors modules/sketch_encoder.py:39-98 and
    returns the per-down-block tuples of residual samples instead."""
    boc = cfg.block_out_channels
    nb = len(boc)
    B = x.shape[0]
    tt = torch.as_tensor(t).reshape(-1).expand(B)
    temb = _r(timestep_embedding(tt, boc[0]).to(x.dtype), "temb")
    temb = _r(F.linear(temb, W["time_embedding.linear_1.weight"], W["time_embedding.linear_1.bias"]), "temb")
    temb = _r(F.linear(_r(F.silu(temb), "temb"), W["time_embedding.linear_2.weight"], W["time_embedding.linear_2.bias"]), "temb")
    temb_act = _r(F.silu(temb), "temb")

    h = _r(F.conv2d(_r(x, "res"), W["conv_in.weight"], W["conv_in.bias"], padding=1), "res")
    skips = [h]
    taps_down, per_block = [], []
    for i in range(nb):
        blk_res = []
        for j in range(cfg.layers_per_block):
            h = resnet_forward(cfg, W, f"down_blocks.{i}.resnets.{j}", h, temb_act)
            if i < nb - 1:
                h = transformer2d_forward(cfg, W, f"down_blocks.{i}.attentions.{j}", h, ehs,
                                          cfg.num_heads[i], inject)
            skips.append(h)
            blk_res.append(h)
        if i < nb - 1:
            p = f"down_blocks.{i}.downsamplers.0.conv"
            h = _r(F.conv2d(_r(h, "rop_dn"), W[p + ".weight"], W[p + ".bias"], stride=2, padding=1), "res")
            skips.append(h)
            blk_res.append(h)
        per_block.append(tuple(blk_res))
        if i < 3:
            taps_down.append(h)
    if down_only:
        return per_block

    h = resnet_forward(cfg, W, "mid_block.resnets.0", h, temb_act)
    tap_mid_r0 = h
    h = transformer2d_forward(cfg, W, "mid_block.attentions.0", h, ehs, cfg.num_heads[-1], inject)
    tap_mid_attn = h
    h = resnet_forward(cfg, W, "mid_block.resnets.1", h, temb_act)
    tap_mid_r1 = h

    taps_up = []
    rev_heads = tuple(reversed(cfg.num_heads))
    for i in range(nb):
        for j in range(cfg.layers_per_block + 1):
            h = torch.cat([h, skips.pop()], dim=1)
            h = resnet_forward(cfg, W, f"up_blocks.{i}.resnets.{j}", h, temb_act)
            if i > 0:
                h = transformer2d_forward(cfg, W, f"up_blocks.{i}.attentions.{j}", h, ehs,
                                          rev_heads[i], inject)
        if i < nb - 1:
            p = f"up_blocks.{i}.upsamplers.0.conv"
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _r(F.conv2d(_r(h, "rop_up"), W[p + ".weight"], W[p + ".bias"], padding=1), "res")
        if i < 3:
            taps_up.append(h)
    h = _r(F.silu(_gn(h, W, "conv_norm_out", cfg.norm_groups, 1e-5)), "norm")
    eps = _r(F.conv2d(h, W["conv_out.weight"], W["conv_out.bias"], padding=1), "res")
    taps = taps_down + [tap_mid_attn, tap_mid_r0, tap_mid_r1] + taps_up
    return eps, taps


def tap_channels(cfg: UNetConfig) -> List[int]:
    b = cfg.block_out_channels
    rev = tuple(reversed(b))
    return [b[0], b[1], b[2], b[-1], b[-1], b[-1], rev[0], rev[1], rev[2]]


def tap_sizes(cfg: UNetConfig, h: int) -> List[int]:
    """Spatial size of each tap for a latent of side h."""
    return [h // 2, h // 4, h // 8, h // 8, h // 8, h // 8, h // 4, h // 2, h]

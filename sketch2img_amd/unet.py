"""SD-style UNet evaluated with the libskg.so kernels: forward (with feature taps) and
backward-to-input.

Replaces what the reference reaches through ``self.unet(...)`` at modules/pipeline.py:96
(diffusers UNet2DConditionModel -> cuDNN / cuBLAS / xformers kernels), the forward hooks of
modules/latent_predictor.py:47-81 (the nine taps) and the autograd pass triggered at
modules/pipeline.py:159.  Weights use the diffusers state_dict key names.

Data layout in HBM: every activation is fp16, token-major [rows*H*W, C]; rows are
ordered [uncond rows of all samples ; cond rows of all samples] so the backward pass (cond rows
only - the uncond gradient is discarded at modules/pipeline.py:159) works on the contiguous second
half of every stashed buffer.  Weight packs: Linear / 1x1 conv [N][K]; conv3x3
[Cout][ky][kx][Cin]; each has a second "dgrad" pack (transposed / tap-flipped) for backward.

Work hoisted out of the per-step loop because it does not depend on the latent:
  * the time-embedding MLP and every ResnetBlock2D.time_emb_proj(silu(temb)) -> one fused bias
    vector per (timestep, resnet), folded into conv1's bias;
  * cross-attention K / V projections of the text embeddings (per prompt).
"""
from __future__ import annotations

import math
import os
from dataclasses import dataclass, field
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

from . import ops
from .config import UNetConfig, up_block_plan

_GN_FROM_PRODUCER = os.environ.get("SKG_GN_PRODUCER", "1") != "0"      # A/B switches (bench.py on one box)
_GEGLU_KEEP = os.environ.get("SKG_GEGLU_KEEP", "1") != "0"
_GN_CONCAT = os.environ.get("SKG_GN_CONCAT", "1") != "0"
_FF_BLOCK = os.environ.get("SKG_FF_BLOCK", "1") != "0"              # fused feed-forward sub-block at C = 320 (csrc/ffblock.hip)
_XATTN_BLOCK = os.environ.get("SKG_XATTN_BLOCK", "1") != "0"        # fused cross-attention sub-block at C = 320, 8 heads (csrc/xattn.hip)
_FF_KEEP = os.environ.get("SKG_FF_KEEP", "1") != "0"                # ... also for the cond rows of a guided step (stashing launch)
_XATTN_HEADS = (8,) if os.environ.get("SKG_XATTN_D64", "1") == "0" else (8, 5)     # head counts of the fused cross-attention launch at C = 320 (5 x 64: round 5)
_XATTN_KEEP = os.environ.get("SKG_XATTN_KEEP", "1") != "0"          # the fused cross-attention launch also in guided steps (stashing launch)
_XATTN_KEEP_HP = os.environ.get("SKG_XATTN_KEEP_HP", "1") != "0"    # ... and in the accuracy mode's guided steps (pairs)
_FF_PROJ = os.environ.get("SKG_FF_PROJ", "1") != "0"                # proj_out + outer residual inside the fused feed-forward launch
_FF_PROJ_HP = os.environ.get("SKG_FF_PROJ_HP", "1") != "0"          # ... in the accuracy mode too (skg_ff_block_proj_f16_hilo, round 5)
_ATTN_DQ_DELTA = os.environ.get("SKG_ATTN_DQ_DELTA", "1") != "0"    # attention backward: delta inside the dQ launch (round 5)
_RES_SC = os.environ.get("SKG_RES_SC", "1") != "0"                  # conv2 + conv_shortcut of a ResnetBlock as one implicit GEMM (round 5)
# round 6: the ResnetBlock convolutions of the two deepest levels (16 x 16 / 8 x 8 at 64 x 64 latents) by Winograd F(2x2, 3x3) (csrc/wino.hip):
# 0 = off, 1 = conv1, conv2 without a shortcut and the data gradients, 2 (default) = also conv2 of the channel-changing blocks of the
# second-deepest level (shortcut GEMM + Winograd with a residual instead of the folded launch: 0.79-0.87 of its time at 16 x 16, 1.17 at
# 8 x 8 - profiles/r06_wino_bench.txt).  A launch takes the path with >= _WINO_MIN_TILES tile positions (output pixels / 4): at 128 - the
# cond-only backward of the 8 x 8 level - the 16 component GEMMs are 128 rows each and the implicit GEMM is as fast.
_WINO = int(os.environ.get("SKG_WINO", "2"))
_WINO_MIN_TILES = int(os.environ.get("SKG_WINO_MIN_TILES", "256"))  # tile positions (output pixels / 4) a launch needs to take the path
_WINO_GN = os.environ.get("SKG_WINO_GN", "1") != "0"                # the GroupNorm in front writes the Winograd input transform itself
# accuracy mode: Winograd also for the ResnetBlock convolutions of the PAIR zone's small maps (the 16 x 16 level with HP_PLAIN_LEVELS = 1):
# pair output through the output transform, the K-doubled shortcut as a pair GEMM whose result is the residual
_HP_WINO = os.environ.get("SKG_HP_WINO", "0") != "0"

CIN_PAD = 64      # latent channels padded to one 64-deep K tile of the LDS-DMA implicit-GEMM conv
COUT_PAD = 8      # conv_out / conv_in-dgrad output channels padded to the 8-channel store granule
CTX_PAD = 8       # text tokens padded to a multiple of 8 (77 -> 80)


def _h(t: torch.Tensor, dev) -> torch.Tensor:
    return t.detach().to(device=dev, dtype=torch.float16).contiguous()


def pack_conv(w: torch.Tensor, dev, cin_pad: int = 0, cout_pad: int = 0) -> torch.Tensor:
    """[Cout,Cin,3,3] -> [Cout(+pad)][ky][kx][Cin(+pad)] flattened to [Cout, 9*Cin]."""
    co, ci = w.shape[:2]
    p = w.permute(0, 2, 3, 1)
    if cin_pad > ci:
        p = torch.nn.functional.pad(p, (0, cin_pad - ci))
    if cout_pad > co:
        p = torch.nn.functional.pad(p, (0, 0, 0, 0, 0, 0, 0, cout_pad - co))
    return _h(p.reshape(p.shape[0], -1), dev)


def pack_conv_wino(w: torch.Tensor, dev, dgrad: bool = False) -> torch.Tensor:
    """Winograd F(2x2, 3x3) weight pack of a 3x3 convolution [Cout, Cin, 3, 3] -> U [Cout, 16 * Cin] fp16: U = G g G^T per (cout, cin),
    formed in fp32 and rounded once; component c = 4 i + j at columns [c Cin, (c + 1) Cin) (ops.conv3x3_wino, csrc/wino.hip).
    dgrad: the pack of the convolution's DATA GRADIENT - itself a 3x3 convolution with the taps flipped and in / out swapped."""
    g = w.detach().to(device=dev, dtype=torch.float32)                   # (formed on the target device, elementwise: no BLAS call at pack time)
    if dgrad:
        g = g.flip(2, 3).transpose(0, 1)

    def G3(a, b, c):                                                     # G = [[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]] applied along one axis
        return a, 0.5 * (a + b + c), 0.5 * (a - b + c), c

    rows = G3(g[:, :, 0, :], g[:, :, 1, :], g[:, :, 2, :])               # G g: 4 x [Cout, Cin, 3]
    U = torch.stack([torch.stack(G3(r[:, :, 0], r[:, :, 1], r[:, :, 2]), 1) for r in rows], 1)      # (G g) G^T: [Cout, 4 (i), 4 (j), Cin]
    return _h(U.reshape(U.shape[0], -1), dev)


def pack_conv_dgrad(w: torch.Tensor, dev, cin_pad: int = 0, cout_pad: int = 0) -> torch.Tensor:
    """dgrad pack: Wd[ci][ky'][kx'][co] = W[co][ci][2-ky'][2-kx'] -> [Cin, 9*Cout].
    ``cin_pad`` pads the OUTPUT rows (the conv's input channels), ``cout_pad`` the contraction."""
    co, ci = w.shape[:2]
    p = w.flip(2, 3).permute(1, 2, 3, 0)
    if cout_pad > co:
        p = torch.nn.functional.pad(p, (0, cout_pad - co))
    if cin_pad > ci:
        p = torch.nn.functional.pad(p, (0, 0, 0, 0, 0, 0, 0, cin_pad - ci))
    return _h(p.reshape(p.shape[0], -1), dev)


def pack_conv_up2(w: torch.Tensor, dev) -> torch.Tensor:
    """Polyphase pack of a 3x3 filter applied after a nearest 2x upsample (ops.conv_up2): [4 phases 2a+b][Cout][4 taps][Cin].
    Output pixel (2i+a, 2j+b) reads low-res rows {i-1, i} (a = 0) or {i, i+1} (a = 1): the filter rows that land on one
    low-res row are summed (fp32, one fp16 rounding), the same for columns."""
    w = w.detach().float()
    co, ci = w.shape[:2]
    phases = []
    for a in (0, 1):
        rws = (w[:, :, 0], w[:, :, 1] + w[:, :, 2]) if a == 0 else (w[:, :, 0] + w[:, :, 1], w[:, :, 2])      # [co, ci, kx]
        for b in (0, 1):
            taps = []
            for r in rws:
                taps += [r[:, :, 0], r[:, :, 1] + r[:, :, 2]] if b == 0 else [r[:, :, 0] + r[:, :, 1], r[:, :, 2]]
            phases.append(torch.stack(taps, 1).reshape(co, 4 * ci))                                            # [co][tap][ci]
    return _h(torch.stack(phases, 0), dev)


def pack_conv_up2_hilo(w: torch.Tensor, dev) -> torch.Tensor:
    """Accuracy-mode polyphase pack (ops.conv_up2_hilo): the pre-summed weights of pack_conv_up2 kept as (hi, lo) fp16 pairs,
    per phase and tap [W_hi | W_hi | W_lo] against the operand blocks [x_hi | x_lo | x_hi]: [4][Cout][4 taps][3 Cin]."""
    w = w.detach().float()
    co, ci = w.shape[:2]
    phases = []
    for a in (0, 1):
        rws = (w[:, :, 0], w[:, :, 1] + w[:, :, 2]) if a == 0 else (w[:, :, 0] + w[:, :, 1], w[:, :, 2])      # [co, ci, kx]
        for b in (0, 1):
            taps = []
            for r in rws:
                taps += [r[:, :, 0], r[:, :, 1] + r[:, :, 2]] if b == 0 else [r[:, :, 0] + r[:, :, 1], r[:, :, 2]]
            t = torch.stack(taps, 1)                                                                           # [co][tap][ci] fp32
            hi = t.half()
            lo = (t - hi.float()).half()
            phases.append(torch.cat([hi, hi, lo], 2).reshape(co, 12 * ci))
    return torch.stack(phases, 0).contiguous().to(dev)


def pack_conv_up2_dgrad(w: torch.Tensor, dev) -> torch.Tensor:
    """Data gradient of the polyphase upsample + conv (ops.conv4x4s2): [Cin][16 taps ky*4+kx][Cout], the transposed pre-summed
    weights of pack_conv_up2.  dX[p] = sum over phases a and taps ty of Wpp[a][ty]^T dY[2 (p - oy(a, ty)) + a] with
    oy(0, .) = (-1, 0), oy(1, .) = (0, +1): rows 2p - 1 .. 2p + 2 of dY, window row ky = 0..3 <-> (a, ty) = (1,1), (0,1), (1,0), (0,0)."""
    w = w.detach().float()
    co, ci = w.shape[:2]

    def split(t, axis):          # 3 filter taps along `axis` -> the four window positions along that axis
        t0, t1, t2 = t.unbind(axis)
        return [t2, t1 + t2, t0 + t1, t0]      # ky = 0: (a=1, ty=1) = w2;  1: (0,1) = w1 + w2;  2: (1,0) = w0 + w1;  3: (0,0) = w0
    taps = []
    for r in split(w, 2):                                        # 4 x [co, ci, kx]
        taps += split(r, 2)                                      # 16 x [co, ci], order ky * 4 + kx
    p = torch.stack(taps, 0)                                     # [16, co, ci]
    return _h(p.permute(2, 0, 1).reshape(ci, 16 * co), dev)


UP2_POLYPHASE = os.environ.get("SKG_UP2_POLY", "1") != "0"      # A/B switch (bench.py on one box)
# accuracy mode: 1 = the upsamplers' K-tripled form ([x_hi | x_lo | x_hi] . [W_hi | W_hi | W_lo]), 0 (default, round 5) = the default
# polyphase launch on x_hi with a pair output: both correction thirds together move eps by 3.6 % of the mode's distance from fp32
# (tools/eps_decompose_up.py) and cost two thirds of the upsamplers' time - the largest single item of the mode's price
HP_UP_TRIPLE = os.environ.get("SKG_HP_UP_TRIPLE", "0") != "0"
# accuracy mode (round 6): GroupNorm OUTPUTS kept as (hi, lo) pairs - the K-doubled operand [n_hi | n_lo] . [W | W] of the matmul that
# consumes them - at the sites whose fp16 rounding carries the most of the mode's remaining eps distance (tools/eps_decompose_sites.py:
# the norm outputs of the LAST up block are 66 % of its variance; conv_norm_out, 18 %, has been a pair since round 4).  Comma-separated
# norm names of the last up block ("<resnet>.norm1" / ".norm2" -> conv1 / conv2, "<attention>.norm" -> proj_in); "" = none (round 5).
# Default = eight of the nine sites (all but norm1 of the first ResnetBlock: the widest operand, 960 channels, for the smallest share),
# chosen together with HP_PLAIN_LEVELS = 1 AT THE REAL BATCH (profiles/r06_eps_real_batch.txt: configs[1]'s 8 samples x 3 timesteps = 48
# rows, ms per 16-row evaluation on one box): round 5's mode rel 5.41e-4 / worst max 1.03e-3 / 18.89 ms; plain 2 + 6 sites + Winograd
# 5.08e-4 / 9.8e-4 / 18.18; plain 1 + 6 sites 4.50e-4 / 8.4e-4 / 19.16; plain 1 + 9 sites 4.18e-4 / 8.1e-4 / 19.66; plain 0 + 9 sites
# 4.14e-4 / 7.7e-4 / 19.88: accuracy costs ~0.5 % of time per 1 % of rel along the whole front - the setting is where the worst row keeps
# ~15 % of north_star's bound (the throughput target has 60 % of headroom, the bound had 2 %).
# ("{last}" = the index of the last up block of the UNet at hand: 3 for SD1.x / SD2.x.)
HP_NORM_PAIRS = tuple(n for n in os.environ.get(
    "SKG_HP_NORM_PAIRS", "up_blocks.{last}.resnets.0.norm2,up_blocks.{last}.resnets.1.norm1,up_blocks.{last}.resnets.1.norm2,"
                         "up_blocks.{last}.resnets.2.norm1,up_blocks.{last}.resnets.2.norm2,up_blocks.{last}.attentions.0.norm,"
                         "up_blocks.{last}.attentions.1.norm,up_blocks.{last}.attentions.2.norm").split(",") if n)
# accuracy mode (round 6): the DEEPEST resolution levels run the default fp16 kernels - tools/eps_decompose_stream.py: of what the pairs
# (residual stream, conv outputs that feed a norm, stream-as-operand) buy, 60 % is bought in the last up block, 25 % in the first down block
# (its skips feed the last up block), 15 % at the 32 x 32 level, ~4 % at the 16 x 16 level and nothing at 8 x 8.  n = number of deepest
# levels (8 x 8, 16 x 16, ...) whose blocks run plain fp16: down_blocks[nb - n ..], mid_block, up_blocks[.. n - 1]; 0 = pairs everywhere.
HP_PLAIN_LEVELS = int(os.environ.get("SKG_HP_PLAIN_LEVELS", "1"))
UP2_SMALL_MAPS = os.environ.get("SKG_UP2_SMALL", "1") != "0"    # A/B: polyphase also where one phase does not fill the chip
UP2_DGRAD = os.environ.get("SKG_UP2_DGRAD", "1") != "0"         # A/B: the upsampler's backward as one 4 x 4 stride-2 convolution


def pack_ff_block(w1: torch.Tensor, b1: torch.Tensor, w2: torch.Tensor, dev, w_proj: Optional[torch.Tensor] = None):
    """Fragment-major pack of a GEGLU feed-forward (diffusers FeedForward: net.0.proj [2F, C] = value rows then gate rows,
    net.2 [C, F]) for skg_ff_block_f16 (csrc/ffblock.hip): every 512-half piece is one MFMA A operand in lane order, so
    the kernel fetches it with one LDS-DMA instruction and reads it with one conflict-free ds_read_b128.
    Chunk c = hidden units 32c .. 32c+31:
      40 W1 pieces (t, ks): t = value 0-15, value 16-31, gate 0-15, gate 16-31; [lane = 16 g + l][i] = W1[row(t, l)][32 ks + 8 g + i]
      20 W2 pieces u:       [lane = 16 g + l][i] = W2[16 u + l][32 c + 16 (i >> 2) + 4 g + (i & 3)]
    (the hidden-unit order of the second product is the accumulator layout of the first).
    w_proj [C, C] (Transformer2DModel.proj_out, skg_ff_block_proj_f16): five more chunks j behind them whose first 40 pieces are
      (t, ks): [lane = 16 g + l][i] = Wp[16 (4 j + t) + l][32 ks + 16 (i >> 2) + 4 g + (i & 3)]   (the block output's accumulator order)
    and whose last 20 pieces are zeros.
    Returns (pack fp16 [F/32 (+ 5), 60, 512], bias1 fp32 [F/32, 4, 16])."""
    F2, C = w1.shape
    F = F2 // 2
    assert w2.shape == (C, F) and C % 32 == 0 and F % 32 == 0
    nch, KS, NU = F // 32, C // 32, C // 16
    # (host-side packing: the state dict may live on the device after a broadcast)
    w1h, w2h, b1 = w1.detach().to("cpu", torch.float16), w2.detach().to("cpu", torch.float16), b1.detach().cpu()
    rows = torch.stack([w1h[:F].reshape(nch, 2, 16, C), w1h[F:].reshape(nch, 2, 16, C)], 1)     # [c, val|gate, half, l, C]
    rows = rows.reshape(nch, 4, 16, KS, 4, 8)                                                     # [c, t, l, ks, g, i]
    p1 = rows.permute(0, 1, 3, 4, 2, 5).reshape(nch, 4 * KS, 512)                                 # [c, (t, ks), (g, l, i)]
    w2r = w2h.reshape(NU, 16, nch, 2, 4, 4)                                                       # [u, l, c, i_hi, g, i_lo]
    p2 = w2r.permute(2, 0, 4, 1, 3, 5).reshape(nch, NU, 512)                                      # [c, u, (g, l, i_hi, i_lo)]
    pack = torch.cat([p1, p2], 1)
    if w_proj is not None:
        assert w_proj.shape == (C, C) and NU % 4 == 0
        wp = w_proj.detach().to("cpu", torch.float16).reshape(NU, 16, KS, 2, 4, 4)             # [u, l, ks, i_hi, g, i_lo]
        pp = wp.permute(0, 2, 4, 1, 3, 5).reshape(NU // 4, 4 * KS, 512)                        # [j, (t, ks), (g, l, i_hi, i_lo)]
        pack = torch.cat([pack, torch.cat([pp, torch.zeros(NU // 4, NU, 512, dtype=torch.float16)], 1)], 0)
    pack = pack.contiguous().to(dev)
    b1h = b1.to(torch.float16).float()
    bias1 = torch.stack([b1h[:F].reshape(nch, 2, 16), b1h[F:].reshape(nch, 2, 16)], 1).reshape(nch, 4, 16).contiguous().to(dev)
    return pack, bias1


def pack_xattn_weights(wq: torch.Tensor, wo: torch.Tensor, heads: int, dev):
    """Fragment-major pack of attn2.to_q [C, C] and attn2.to_out.0 [C, C] for skg_xattn_block_f16 (csrc/xattn.hip), C = 320; per head h
    the kernel's LDS image in pieces of 512 halves.
    8 heads of 40 (SD1.5; head width padded to 48), 60 pieces:
      30 Wq pieces (t, ks):  [lane = 16 g + l][i] = Wq[40 h + 16 t + l][32 ks + 8 g + i]         (rows 40..47 of the head: zeros)
      Wo image (30 pieces):  20 K = 32 fragments  [lane][i] = Wo[16 u + l][40 h + 16 (i >> 2) + 4 g + (i & 3)]
                             20 K = 16 fragments  [lane][i < 4] = Wo[16 u + l][40 h + 32 + 4 g + i]   (d >= 40: zeros)
    5 heads of 64 (SD2.1), 80 pieces: 40 Wq pieces (t < 4, ks) as above, then 2 x 20 K = 32 fragments
      [s][u]: [lane][i] = Wo[16 u + l][64 h + 32 s + 16 (i >> 2) + 4 g + (i & 3)].
    Returns fp16 [heads, 60 | 80, 512]."""
    C = wq.shape[0]
    dh = C // heads
    assert wq.shape == (C, C) and wo.shape == (C, C) and C == 320 and dh in (40, 64)
    KS, NU = C // 32, C // 16
    DP = 48 if dh == 40 else 64
    NT = DP // 16
    wqh, woh = wq.detach().to("cpu", torch.float16), wo.detach().to("cpu", torch.float16)      # host-side packing
    out = []
    for h in range(heads):
        q = torch.zeros(DP, C, dtype=torch.float16)
        q[:dh] = wqh[h * dh:(h + 1) * dh]
        pq = q.reshape(NT, 16, KS, 4, 8).permute(0, 2, 3, 1, 4).reshape(NT * KS, 512)          # [t, ks][g, l, i]
        o = torch.zeros(C, DP, dtype=torch.float16)
        o[:, :dh] = woh[:, h * dh:(h + 1) * dh]
        o32 = o[:, :32].reshape(NU, 16, 2, 4, 4).permute(0, 3, 1, 2, 4).reshape(-1)             # [u][g, l, i_hi, i_lo]
        if dh == 40:
            tail = o[:, 32:].reshape(NU, 16, 4, 4).permute(0, 2, 1, 3).reshape(-1)              # [u][g, l, i]
        else:
            tail = o[:, 32:].reshape(NU, 16, 2, 4, 4).permute(0, 3, 1, 2, 4).reshape(-1)        # the second K = 32 step: d 32..63
        out.append(torch.cat([pq.reshape(-1), o32, tail]).reshape(-1, 512))
    return torch.stack(out).contiguous().to(dev)


def pack_xattn_kv(K: torch.Tensor, V: torch.Tensor, rows: int, Lp: int, L: int, heads: int) -> torch.Tensor:
    """Fragment-major pack of the text keys / values of every batch row for skg_xattn_block_f16: K, V [rows * Lp, C] (what
    prepare_context hoists per prompt), L <= 80 valid keys per row.  Per (row, head), pieces of 512 halves.
    Head width 40 (16 pieces):
      K image: 5 K = 32 fragments [lane = 16 g + l][i] = K[key 16 kt + l][40 h + 16 (i >> 2) + 4 g + (i & 3)], then 5 K = 16
               fragments [lane][i < 4] = K[key 16 kt + l][40 h + 32 + 4 g + i]                                   (8 pieces)
      V image: (dt, s) K = 32 fragments [lane][i] = V[key 32 s + 16 (i >> 2) + 4 g + (i & 3)][40 h + 16 dt + l], then 3 K = 16
               fragments [lane][i < 4] = V[key 64 + 4 g + i][40 h + 16 dt + l]                                   (8 pieces)
    Head width 64 (20 pieces): K image [s][kt] 2 x 5 K = 32 fragments (d = 64 h + 32 s + ...), V image (dt < 4, s) 8 K = 32 fragments
    then 4 K = 16 fragments (2 pieces); no padding.
    Keys >= L and head columns >= the head width are zeros.  Returns fp16 [rows, heads, 16 | 20, 512] on K's device."""
    C = K.shape[1]
    dh = C // heads
    assert dh in (40, 64) and L <= 80 and K.shape == V.shape == (rows * Lp, C)
    dev = K.device
    DP = 48 if dh == 40 else 64
    NT = DP // 16
    k = torch.zeros(rows, heads, 80, DP, device=dev, dtype=torch.float16)
    v = torch.zeros(rows, heads, 80, DP, device=dev, dtype=torch.float16)
    k[:, :, :L, :dh] = K.reshape(rows, Lp, heads, dh)[:, :L].permute(0, 2, 1, 3)
    v[:, :, :L, :dh] = V.reshape(rows, Lp, heads, dh)[:, :L].permute(0, 2, 1, 3)
    R = rows * heads
    k, v = k.reshape(R, 80, DP), v.reshape(R, 80, DP)
    k32 = k[:, :, :32].reshape(R, 5, 16, 2, 4, 4).permute(0, 1, 4, 2, 3, 5).reshape(R, 5 * 512)      # [kt][g, l, i_hi, i_lo]
    v32 = v[:, :64].reshape(R, 2, 2, 4, 4, NT, 16).permute(0, 5, 1, 3, 6, 2, 4).reshape(R, 2 * NT * 512)   # [dt, s][g, l, i_hi, i_lo]
    v16 = v[:, 64:].reshape(R, 4, 4, NT, 16).permute(0, 3, 1, 4, 2).reshape(R, NT * 256)             # [dt][g, l, i]
    if dh == 64:
        k32b = k[:, :, 32:].reshape(R, 5, 16, 2, 4, 4).permute(0, 1, 4, 2, 3, 5).reshape(R, 5 * 512)
        return torch.cat([k32, k32b, v32, v16], 1).reshape(rows, heads, 20, 512).contiguous()
    k16 = k[:, :, 32:].reshape(R, 5, 16, 4, 4).permute(0, 1, 3, 2, 4).reshape(R, 5 * 256)            # [kt][g, l, i]
    pad_k = torch.zeros(R, 4096 - 5 * 768, device=dev, dtype=torch.float16)
    pad_v = torch.zeros(R, 4096 - 6 * 512 - 3 * 256, device=dev, dtype=torch.float16)
    return torch.cat([k32, k16, pad_k, v32, v16, pad_v], 1).reshape(rows, heads, 16, 512).contiguous()


def _pad_vec(v: torch.Tensor, n: int) -> torch.Tensor:
    return torch.nn.functional.pad(v, (0, n - v.shape[0])) if n > v.shape[0] else v


@dataclass
class Stash:
    """Activations kept by a grad-enabled forward for the backward pass."""
    res: Dict[str, dict] = field(default_factory=dict)
    tr: Dict[str, dict] = field(default_factory=dict)
    misc: dict = field(default_factory=dict)


class _Packs(dict):
    """Weight packs by name.  `lazy[k] = make` registers a pack that is only built (from a host copy of the tensor) the first time it
    is read: the packs of code paths that normally never run - the two-launch fall-back of a declined fused launch (they
    were ~0.5 GB of resident duplicates)."""

    def __init__(self):
        super().__init__()
        self.lazy: Dict[str, Callable[[], torch.Tensor]] = {}

    def __missing__(self, k):
        v = self[k] = self.lazy.pop(k)()         # (KeyError for a name that was never registered)
        return v

    def __contains__(self, k):
        return dict.__contains__(self, k) or k in self.lazy


class HipUNet:
    def __init__(self, cfg: UNetConfig, state_dict: Dict[str, torch.Tensor], device="cuda",
                 need_backward: bool = True, residual_fp32: bool = False):
        """residual_fp32: opt-in ACCURACY mode that meets north_star's <= 1e-3 max eps deviation from the fp32 reference: the
        residual stream and every conv output that feeds a norm or the residual sum are kept as (hi, lo) pairs of fp16 tensors
        (~22 mantissa bits), see forward() / _forward_hp().  Where the stream itself is a matmul operand the pair is the K-doubled
        operand (shortcuts, downsamplers, proj_out, conv_out) - EXCEPT the three nearest-2x upsampler convolutions, which read the hi
        part only and write a pair (round 5: SKG_HP_UP_TRIPLE=1 restores their pair operand; the declined-launch fall-back keeps the
        hi-only operand too).  Guided steps work: the pair forward stashes the hi activations and the ordinary backward runs on them."""
        self.cfg = cfg
        self.dev = torch.device(device)
        self.need_backward = need_backward
        self.residual_fp32 = residual_fp32
        self.W: Dict[str, torch.Tensor] = _Packs()
        self.hp_norm_pairs: set = set()
        nb = len(cfg.block_out_channels)
        self.hp_plain_levels = max(0, min(HP_PLAIN_LEVELS, nb - 1)) if residual_fp32 else 0
        self._pack(state_dict)
        if residual_fp32:
            assert all(c % 64 == 0 for c in cfg.block_out_channels), \
                "accuracy mode: the pair epilogue lives in the LDS-DMA kernels (channel counts must be multiples of 64)"
            self._pack_hp(state_dict)
        self._sd_time = {k: v for k, v in state_dict.items()
                         if k.startswith("time_embedding.") or ".time_emb_proj." in k or k.endswith("conv1.bias")}
        self.tbias: Dict[int, Dict[str, torch.Tensor]] = {}
        self.ctx: Optional[dict] = None
        self.inject: Optional[Callable] = None      # set by modules.*_guided_attn.SatMixin

    def _hp_plain(self, name: str) -> bool:
        """Accuracy mode: does this module (state-dict prefix) run the default fp16 kernels (HP_PLAIN_LEVELS)?  Resolution level of
        down_blocks.i = i, of mid_block = nb - 1, of up_blocks.i = nb - 1 - i; plain iff level >= nb - hp_plain_levels."""
        if not self.residual_fp32 or self.hp_plain_levels == 0:
            return not self.residual_fp32
        nb, pl = len(self.cfg.block_out_channels), self.hp_plain_levels
        part = name.split(".")
        if part[0] == "down_blocks":
            return int(part[1]) >= nb - pl
        if part[0] == "up_blocks":
            return int(part[1]) <= pl - 1
        return part[0] == "mid_block"

    # ------------------------------------------------------------------ packing
    def _pack(self, sd):
        W, dev, bw = self.W, self.dev, self.need_backward
        # ResnetBlocks whose conv2 + conv_shortcut run as ONE implicit GEMM (ops.conv3x3_sc): the forward packs of the two separate
        # launches are only read when that launch is declined (an operand >= 2 GiB) - registered lazily, from fp16 host copies
        folded = set()
        if _RES_SC:
            for k, v in sd.items():
                if k.endswith(".conv_shortcut.weight"):
                    r = k[: -len(".conv_shortcut.weight")]
                    if sd[r + ".conv2.weight"].shape[1] % 64 == 0 and v.shape[1] % 64 == 0:
                        folded.add(r)

        def lazy(key, tensor, make):
            host = tensor.detach().to("cpu", torch.float16)
            W.lazy[key] = lambda: make(host)

        for k, v in sd.items():
            r = k.rsplit(".", 2)[0]
            if r in folded and k.endswith(".conv2.weight"):
                lazy(k, v, lambda h: pack_conv(h, dev))
                if bw:
                    W[k + ":T"] = pack_conv_dgrad(v, dev)
                continue
            if r in folded and k.endswith(".conv_shortcut.weight"):
                lazy(k, v, lambda h: _h(h.reshape(h.shape[0], h.shape[1]), dev))
                if bw:
                    W[k + ":T"] = _h(v.reshape(v.shape[0], v.shape[1]).t(), dev)
                continue
            if v.dim() == 4 and v.shape[2] == 3:
                if k == "conv_in.weight":
                    W[k] = pack_conv(v, dev, cin_pad=CIN_PAD)
                    if bw:
                        W[k + ":T"] = pack_conv_dgrad(v, dev, cin_pad=COUT_PAD)
                elif k == "conv_out.weight":
                    W[k] = pack_conv(v, dev, cout_pad=COUT_PAD)
                else:
                    W[k] = pack_conv(v, dev)
                    if bw:
                        W[k + ":T"] = pack_conv_dgrad(v, dev)
                    if ".upsamplers." in k and UP2_POLYPHASE and v.shape[1] % 64 == 0:
                        W[k + ":pp"] = pack_conv_up2(v, dev)
                        if bw and v.shape[0] % 64 == 0:
                            W[k + ":ppT"] = pack_conv_up2_dgrad(v, dev)
            elif v.dim() == 4:                                  # 1x1 conv
                W[k] = _h(v.reshape(v.shape[0], v.shape[1]), dev)
                if bw:
                    W[k + ":T"] = _h(v.reshape(v.shape[0], v.shape[1]).t(), dev)
            elif v.dim() == 2:
                if ".attn1.to_" in k and not k.endswith("to_out.0.weight"):
                    continue                                    # fused below
                if ".attn2.to_k." in k or ".attn2.to_v." in k:
                    W[k] = _h(v, dev)                            # used once per prompt
                    continue
                if ".time_emb" in k:
                    W[k] = _h(v, dev)
                    continue
                W[k] = _h(v, dev)
                if bw:
                    W[k + ":T"] = _h(v.t(), dev)
            else:
                W[k] = _h(_pad_vec(v, COUT_PAD) if k == "conv_out.bias" else v, dev)
        # conv2 + conv_shortcut as ONE implicit GEMM (ops.conv3x3_sc): [conv2 tap-major pack | W_sc] along K, biases summed
        if _WINO and len(self.cfg.block_out_channels) >= 3:
            nb = len(self.cfg.block_out_channels)
            for blk, deepest in ((f"down_blocks.{nb - 2}.", False), ("up_blocks.1.", False), (f"down_blocks.{nb - 1}.", True),
                                 ("mid_block.", True), ("up_blocks.0.", True)):
                for k, v in sd.items():
                    if k.startswith(blk) and ".resnets." in k and (k.endswith(".conv1.weight") or k.endswith(".conv2.weight")) and v.shape[1] % 64 == 0:
                        if deepest and k.endswith(".conv2.weight") and (k[: -len(".conv2.weight")] + ".conv_shortcut.weight") in sd:
                            continue      # (8 x 8: the folded shortcut launch stays ahead of shortcut GEMM + Winograd)
                        if self._hp_plain(k) or _HP_WINO:      # (accuracy mode: the forward of a pair-zone block takes Winograd only with SKG_HP_WINO)
                            W[k + ":wino"] = pack_conv_wino(v, dev)
                        if bw and v.shape[0] % 64 == 0 and not deepest:      # (the deepest level's cond-only backward stays on the implicit GEMM)
                            W[k + ":winoT"] = pack_conv_wino(v, dev, dgrad=True)
        for r in sorted(folded):
            w2, wsc = sd[r + ".conv2.weight"], sd[r + ".conv_shortcut.weight"]
            wsc = wsc.reshape(wsc.shape[0], wsc.shape[1])
            w2p = w2.permute(0, 2, 3, 1).reshape(w2.shape[0], -1)                   # [Cout][ky][kx][Cin] (pack_conv's order)
            if not self._hp_plain(r):   # (the accuracy mode's pair blocks read only :sc2 - the pair operand [x_hi | x_lo] . [W | W] - and the summed bias)
                W[r + ".conv2.weight:sc2"] = _h(torch.cat([w2p, wsc, wsc], 1), dev)
            else:
                W[r + ".conv2.weight:sc"] = _h(torch.cat([w2p, wsc], 1), dev)
            W[r + ".conv2.bias:sc"] = _h(sd[r + ".conv2.bias"].float() + sd[r + ".conv_shortcut.bias"].float(), dev)
        # FF1 (GEGLU projection): rows interleaved [a a g g] so the GEMM epilogue can gate in registers
        for k in list(sd.keys()):
            if k.endswith(".ff.net.0.proj.weight"):
                idx = ops.geglu_interleave_index(sd[k].shape[0] // 2)
                W[k] = _h(sd[k][idx], dev)
                W[k[:-len("weight")] + "bias"] = _h(sd[k[:-len("weight")] + "bias"][idx], dev)
                if bw:
                    W[k + ":T"] = _h(sd[k][idx].t(), dev)
        # first level (C = 320; SD1.5: 8 heads of 40, SD2.1: 5 heads of 64): norm2 -> to_q -> text attention -> to_out + residual as
        # ONE row-local launch (csrc/xattn.hip); the per-prompt K / V packs are made in prepare_context
        cfg = self.cfg
        heads320 = cfg.num_heads[list(cfg.block_out_channels).index(320)] if 320 in cfg.block_out_channels else None
        for k in list(sd.keys()):
            if k.endswith(".attn2.to_q.weight") and sd[k].shape == (320, 320) and heads320 in _XATTN_HEADS:
                t = k[: -len(".to_q.weight")]
                W[t + ".xpack"] = pack_xattn_weights(sd[k], sd[t + ".to_out.0.weight"], heads320, dev)
        # 64 x 64 level (C = 320): the whole feed-forward sub-block as ONE row-local launch (csrc/ffblock.hip)
        for k in list(sd.keys()):
            if k.endswith(".ff.net.0.proj.weight") and sd[k].shape[1] == 320 and sd[k].shape[0] // 2 <= 1280:
                t = k[: -len(".ff.net.0.proj.weight")]
                # (with Transformer2DModel.proj_out behind it when the block has one of the same width: skg_ff_block_proj_f16;
                # the plain launch reads the leading chunks of the same tensor)
                tp = t[: -len(".transformer_blocks.0")] if t.endswith(".transformer_blocks.0") else None
                wpj = sd.get(tp + ".proj_out.weight") if tp else None
                wpj = wpj.reshape(wpj.shape[0], -1) if wpj is not None and wpj.numel() == 320 * 320 else None
                packp, W[t + ".ff.bias1"] = pack_ff_block(sd[k], sd[t + ".ff.net.0.proj.bias"], sd[t + ".ff.net.2.weight"], dev, w_proj=wpj)
                nchf = sd[k].shape[0] // 64
                W[t + ".ff.pack"] = packp[:nchf]
                if wpj is not None:
                    W[t + ".ff.packp"] = packp
        for k in list(sd.keys()):
            if k.endswith(".attn1.to_q.weight"):
                p = k[: -len(".to_q.weight")]
                qkv = torch.cat([sd[p + ".to_q.weight"], sd[p + ".to_k.weight"], sd[p + ".to_v.weight"]], 0)
                W[p + ".qkv"] = _h(qkv, dev)
                if bw:
                    W[p + ".qkv:T"] = _h(qkv.t(), dev)

    def _pack_hp(self, sd):
        """Accuracy mode: where the residual stream itself is a matmul operand (conv_shortcut, the resampling
        convolutions, proj_out) the operand is the PAIR [hi | lo] along K and the weight pack is [W | W]."""
        W, dev = self.W, self.dev
        for k, v in sd.items():
            if self._hp_plain(k):       # (a block of the deepest levels: runs the default kernels, HP_PLAIN_LEVELS)
                continue
            if k.endswith(".conv_shortcut.weight") and (k[: -len(".conv_shortcut.weight")] + ".conv2.weight:sc2") in W:
                host = v.detach().to("cpu", torch.float16)          # (fall-back of a declined conv3x3_sc launch only)
                W.lazy[k + ":2"] = lambda h=host: _h(torch.cat([h.reshape(h.shape[0], h.shape[1])] * 2, 1), dev)
            elif k.endswith(".conv_shortcut.weight") or k.endswith(".proj_out.weight"):
                w = v.reshape(v.shape[0], v.shape[1])
                W[k + ":2"] = _h(torch.cat([w, w], 1), dev)
            elif ".downsamplers." in k and k.endswith(".weight") or ".upsamplers." in k and k.endswith(".weight"):
                if ".upsamplers." in k and UP2_POLYPHASE and v.shape[1] % 64 == 0:
                    if getattr(self, "_sd_cpu", None) is None:
                        self._sd_cpu = {}
                    self._sd_cpu[k] = v.detach().to("cpu", torch.float16)      # (source of the 9-tap fall-back pack: _w9x2)
                    if HP_UP_TRIPLE or (k + ":pp") not in W:
                        W[k + ":pp3"] = pack_conv_up2_hilo(v, dev)  # polyphase with (hi, lo) pre-summed weights, K tripled
                    # (else: the default polyphase pack W[k + ":pp"] on x_hi, pair output only)
                else:
                    W[k + ":2"] = pack_conv(torch.cat([v, v], 1), dev)
            elif k == "conv_out.weight":      # its operand - the last normalised activation - reaches eps one to one: a pair too
                W[k + ":2"] = pack_conv(torch.cat([v, v], 1), dev, cout_pad=COUT_PAD)
        # norm outputs as pairs (HP_NORM_PAIRS): the [W | W] pack of the matmul behind each listed norm
        self.hp_norm_pairs = set()
        for name in HP_NORM_PAIRS:
            name = name.replace("{last}", str(len(self.cfg.block_out_channels) - 1))
            r, which = name.rsplit(".", 1)
            if which == "norm" and (r + ".proj_in.weight") in sd:
                w = sd[r + ".proj_in.weight"]
                w = w.reshape(w.shape[0], w.shape[1])
                W[r + ".proj_in.weight:n2"] = _h(torch.cat([w, w], 1), dev)
            elif which == "norm1" and (r + ".conv1.weight") in sd:
                W[r + ".conv1.weight:n2"] = pack_conv(torch.cat([sd[r + ".conv1.weight"]] * 2, 1), dev)
            elif which == "norm2" and (r + ".conv2.weight") in sd:
                w2 = sd[r + ".conv2.weight"]
                w2p = torch.cat([w2, w2], 1).permute(0, 2, 3, 1).reshape(w2.shape[0], -1)      # per tap [W | W]
                if (r + ".conv2.weight:sc2") in W:      # folded shortcut: [conv2 taps on the pair | W_sc | W_sc]
                    wsc = sd[r + ".conv_shortcut.weight"]
                    wsc = wsc.reshape(wsc.shape[0], wsc.shape[1])
                    W[r + ".conv2.weight:n2"] = _h(torch.cat([w2p, wsc, wsc], 1), dev)
                else:
                    W[r + ".conv2.weight:n2"] = _h(w2p, dev)
            elif "SKG_HP_NORM_PAIRS" in os.environ:
                raise ValueError(f"SKG_HP_NORM_PAIRS: {name!r} is not a GroupNorm of this UNet")
            else:
                continue                   # (a default site this architecture does not have: fewer layers per block)
            self.hp_norm_pairs.add(name)

    def _w9x2(self, k: str) -> torch.Tensor:
        """The 9-tap [W | W] pack of an upsampler (accuracy mode), built on first use: only a declined polyphase launch needs it."""
        if k + ":2" not in self.W:
            v = self._sd_cpu[k] if getattr(self, "_sd_cpu", None) is not None else None
            if v is None:
                raise RuntimeError(f"{k}: the polyphase launch was declined and the layer's 9-tap pack is not available")
            self.W[k + ":2"] = pack_conv(torch.cat([v, v], 1), self.dev)
        return self.W[k + ":2"]

    # ------------------------------------------------------------------ hoisted precompute
    def prepare_timesteps(self, timesteps: Sequence[int]):
        """Time-embedding MLP and the per-resnet time projections for every timestep of the schedule
        (diffusers get_timestep_embedding(flip_sin_to_cos=True, shift 0) -> TimestepEmbedding ->
        ResnetBlock2D.time_emb_proj(silu(.))), folded with conv1.bias."""
        cfg, W = self.cfg, self.W
        ts = [int(t) for t in timesteps if int(t) not in self.tbias]
        if not ts:
            return
        c0 = cfg.block_out_channels[0]
        half = c0 // 2
        expo = -math.log(10000.0) * torch.arange(half, dtype=torch.float32) / half
        emb = torch.tensor(ts, dtype=torch.float32)[:, None] * torch.exp(expo)[None, :]
        emb = torch.cat([torch.cos(emb), torch.sin(emb)], dim=-1)              # host table, fp32
        e = emb.to(self.dev, torch.float16).contiguous()
        e = ops.gemm(e, W["time_embedding.linear_1.weight"], bias=W["time_embedding.linear_1.bias"])
        e = ops.silu(e)
        e = ops.gemm(e, W["time_embedding.linear_2.weight"], bias=W["time_embedding.linear_2.bias"])
        e = ops.silu(e)                                                          # silu(temb), [T, 4*c0]
        out = {t: {} for t in ts}
        for k in W:
            if k.endswith(".time_emb_proj.weight"):
                p = k[: -len(".time_emb_proj.weight")]
                proj = ops.gemm(e, W[k], bias=W[p + ".time_emb_proj.bias"])       # [T, Cout]
                fused = ops.axpby(proj, W[p + ".conv1.bias"].expand(len(ts), -1).contiguous())
                for i, t in enumerate(ts):
                    out[t][p] = fused[i].contiguous()
        self.tbias.update(out)

    def prepare_context(self, ehs: torch.Tensor):
        """Cross-attention K / V of the text embeddings, per transformer block.  ehs [rows, L, D] in the
        row order [uncond rows; cond rows]."""
        cfg, W = self.cfg, self.W
        rows, L, D = ehs.shape
        Lp = (L + CTX_PAD - 1) // CTX_PAD * CTX_PAD
        Dp = (D + 31) // 32 * 32
        x = torch.zeros(rows, Lp, Dp, device=self.dev, dtype=torch.float16)
        x[:, :L, :D] = ehs.to(self.dev, torch.float16)
        x = x.reshape(rows * Lp, Dp)
        S = rows // 2
        ctx = dict(L=L, Lp=Lp, rows=rows, blocks={})
        for k in W:
            if k.endswith(".attn2.to_k.weight"):
                p = k[: -len(".to_k.weight")]
                wk, wv = W[k], W[p + ".to_v.weight"]
                if Dp != D:
                    wk = torch.nn.functional.pad(wk, (0, Dp - D)).contiguous()
                    wv = torch.nn.functional.pad(wv, (0, Dp - D)).contiguous()
                Kc = ops.gemm(x, wk)                     # padded token rows are exactly zero (zero input, no bias)
                Vc = ops.gemm(x, wv)
                ctx["blocks"][p] = dict(K=Kc, V=Vc)
                if (p + ".xpack") in W and L <= 80:
                    ctx["blocks"][p]["kvpack"] = pack_xattn_kv(Kc, Vc, rows, Lp, L, W[p + ".xpack"].shape[0])
        self.ctx = ctx

    # ------------------------------------------------------------------ modules, forward
    def _gn_from_producer(self, rows: int, HW: int, C: int) -> bool:
        """GroupNorm inputs of the 64 x 64 / 32 x 32 levels get their statistics from the epilogue of the kernel that
        writes them (ops.gemm / ops.conv3x3 with gn_stats= / gn_groups=): the separate statistics pass - a second read of
        a 10-40 MB tensor - disappears.  Smaller maps keep the one-launch GroupNorm that holds a slice in registers."""
        return _GN_FROM_PRODUCER and HW >= 1024 and ops.gn_fusable(rows * HW, C, HW, self.cfg.norm_groups)

    def _wino_ok(self, key: str, rows: int, H: int) -> bool:
        return key in self.W and rows * H * H // 4 >= _WINO_MIN_TILES and not (H & 1)

    def _conv_wino(self, key: str, x, rows: int, H: int, **kw):
        """The 3x3 convolution `key` (a weight name + ':wino' / ':winoT') by Winograd F(2x2, 3x3) when the pack exists and the launch is
        large enough; None when the path does not take it (the caller runs the implicit GEMM)."""
        if not self._wino_ok(key, rows, H):
            return None
        try:
            return ops.conv3x3_wino(x, self.W[key], rows, H, H, **kw)
        except ops.SkgError as e:      # declined (no room for the slabs in the stream's workspace)
            if e.rc != -2:
                raise
            return None

    def _gn_conv_wino(self, nkey: str, wkey: str, x, rows: int, H: int, eps: float, **kw):
        """GroupNorm + SiLU `nkey` and the Winograd convolution `wkey` behind it with the norm writing the convolution's input transform
        (ops.groupnorm_wino: no normalised tensor in memory).  -> (conv output, statistics), or None when either step declines (nothing
        was consumed: the caller runs groupnorm and the convolution as before)."""
        if not _WINO_GN or not self._wino_ok(wkey, rows, H):
            return None
        W = self.W
        try:
            V, st = ops.groupnorm_wino(x, rows, H, H, self.cfg.norm_groups, eps, W[nkey + ".weight"], W[nkey + ".bias"], True)
            return ops.conv3x3_wino(None, W[wkey], rows, H, H, V=V, **kw), st
        except ops.SkgError as e:
            if e.rc != -2:
                raise
            return None

    def _res_fwd(self, p, x, rows, H, tb, stash: Optional[Stash], out=None, xpart=None, want_part=False, half=False):
        """xpart: GroupNorm partial sums of x from its producer (or None).  Returns (out, partial sums of out or None):
        they are produced when want_part is set and the level takes its statistics from the producers.
        half: `rows` are the COND rows only (the shared CFG prefix, see forward): the stash says so."""
        cfg, W = self.cfg, self.W
        G, HW = cfg.norm_groups, H * H
        Cout = W[p + ".conv1.weight"].shape[0]
        fuse = self._gn_from_producer(rows, HW, Cout)
        # small maps: conv1 / conv2 by Winograd F(2x2, 3x3), the GroupNorm in front writing the input transform where its slice fits a workgroup
        h1 = st1 = None
        if not fuse and xpart is None:
            got = self._gn_conv_wino(p + ".norm1", p + ".conv1.weight:wino", x, rows, H, 1e-5, bias=tb[p])
            if got is not None:
                h1, st1 = got
        if h1 is None:
            n1, st1 = ops.groupnorm(x, rows, HW, G, 1e-5, W[p + ".norm1.weight"], W[p + ".norm1.bias"], True, partial=xpart)
            h1 = None if fuse else self._conv_wino(p + ".conv1.weight:wino", n1, rows, H, bias=tb[p])
        if h1 is not None:
            part1 = None
        elif fuse:
            h1, part1 = ops.conv3x3(n1, W[p + ".conv1.weight"], rows, H, H, bias=tb[p], gn_groups=G)
        else:
            h1, part1 = ops.conv3x3(n1, W[p + ".conv1.weight"], rows, H, H, bias=tb[p]), None
        opart = None
        done = False
        has_sc = (p + ".conv_shortcut.weight") in W
        wino2 = (self._wino_ok(p + ".conv2.weight:wino", rows, H) and not (want_part and fuse) and (not has_sc or _WINO >= 2))
        n2 = st2 = None
        if wino2:
            # Winograd conv2 with the block input - or the shortcut GEMM's output - as its residual
            sc = ops.gemm(x, W[p + ".conv_shortcut.weight"], bias=W[p + ".conv_shortcut.bias"]) if has_sc else x
            got = self._gn_conv_wino(p + ".norm2", p + ".conv2.weight:wino", h1, rows, H, 1e-5, out=out, bias=W[p + ".conv2.bias"], residual=sc)
            if got is not None:
                (out, st2), done = got, True
        if not done:
            n2, st2 = ops.groupnorm(h1, rows, HW, G, 1e-5, W[p + ".norm2.weight"], W[p + ".norm2.bias"], True, partial=part1)
        if wino2 and not done:
            o = self._conv_wino(p + ".conv2.weight:wino", n2, rows, H, out=out, bias=W[p + ".conv2.bias"], residual=sc)
            if o is not None:
                out, done = o, True
        if not done and (p + ".conv2.weight:sc") in W:
            # the 1x1 shortcut as K tiles behind conv2's 3x3 walk: one launch, no [M, Cout] round trip of the shortcut output
            try:
                if want_part and fuse:
                    out, opart = ops.conv3x3_sc(n2, x, W[p + ".conv2.weight:sc"], rows, H, H, out=out, bias=W[p + ".conv2.bias:sc"], gn_groups=G)
                else:
                    out = ops.conv3x3_sc(n2, x, W[p + ".conv2.weight:sc"], rows, H, H, out=out, bias=W[p + ".conv2.bias:sc"])
                done = True
            except ops.SkgError as e:      # declined (an operand >= 2 GiB): the two launches
                if e.rc != -2:
                    raise
        if not done:
            if (p + ".conv_shortcut.weight") in W:
                sc = ops.gemm(x, W[p + ".conv_shortcut.weight"], bias=W[p + ".conv_shortcut.bias"])
            else:
                sc = x
            if want_part and fuse:
                out, opart = ops.conv3x3(n2, W[p + ".conv2.weight"], rows, H, H, out=out, bias=W[p + ".conv2.bias"],
                                         residual=sc, gn_groups=G)
            else:
                out = ops.conv3x3(n2, W[p + ".conv2.weight"], rows, H, H, out=out, bias=W[p + ".conv2.bias"], residual=sc)
        if stash is not None:
            stash.res[p] = dict(x=x, st1=st1, h1=h1, st2=st2, H=H, half=half)
        return out, opart

    def _tr_fwd(self, p, x, rows, H, heads, stash: Optional[Stash], out=None, xpart=None, want_part=False, shared=False,
                x_full=None):
        """shared: x holds the COND rows only (rows // 2 of them) of a CFG-doubled batch whose two halves are identical up
        to here (see forward): everything in front of the first text-dependent operation - GroupNorm, proj_in, the whole
        self-attention, LayerNorm 2, to_q - runs once; p1 and q2 are written into the cond half of full-size buffers and
        copied to the uncond half (x_full: the same for x, filled by the caller)."""
        cfg, W = self.cfg, self.W
        HW = H * H
        C = x.shape[1]
        dh = C // heads
        scale = dh ** -0.5
        t = p + ".transformer_blocks.0"
        keep = stash is not None
        r1 = rows // 2 if shared else rows                    # rows of the text-independent part
        g, gst = ops.groupnorm(x, r1, HW, cfg.norm_groups, 1e-6, W[p + ".norm.weight"], W[p + ".norm.bias"], False,
                               partial=xpart)
        pin = ops.gemm(g, W[p + ".proj_in.weight"], bias=W[p + ".proj_in.bias"])
        a1, st1 = ops.layernorm(pin, W[t + ".norm1.weight"], W[t + ".norm1.bias"], want_stats=True)
        qkv = ops.gemm(a1, W[t + ".attn1.qkv"])
        # V straight out of the fused projection (row-major; the kernel's LDS transpose read replaces the V^T copy)
        o1, lse1 = ops.attn_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], r1, heads, HW, HW, HW, dh, scale,
                                want_lse=True, v_rows=True)
        M1 = r1 * HW
        p1_full = torch.empty(rows * HW, C, device=x.device, dtype=torch.float16) if shared else None
        p1 = ops.gemm(o1, W[t + ".attn1.to_out.0.weight"], p1_full[M1:] if shared else None,
                      bias=W[t + ".attn1.to_out.0.bias"], residual=pin)
        x_c, p1_c = x, p1                                     # what the backward of the cond rows reads
        if self.inject is not None:
            if shared and getattr(self.inject, "halves_equal", False):
                # the injected K / V are the same for both halves too (checked by the injector): stay shared through it
                p1_full2 = torch.empty(rows * HW, C, device=x.device, dtype=torch.float16)
                p1 = self.inject(t, p1, rows, HW, heads, cond_only=True, out=p1_full2[M1:])
                p1_full = p1_full2
            else:
                if shared:                                    # the injected K / V differ between the halves: diverge here
                    ops.batch_copy(p1, M1, p1_full, M1, 1, M1)
                    x, p1, shared = x_full, p1_full, False
                p1 = self.inject(t, p1, rows, HW, heads)
            p1_c = p1
        cb = self.ctx["blocks"][t + ".attn2"]
        xab = (_XATTN_BLOCK and (not keep or (_XATTN_KEEP and rows % 2 == 0)) and "kvpack" in cb and heads in _XATTN_HEADS and HW % 128 == 0)
        xk_half = ()
        if xab:
            # norm2 -> to_q -> attention over the text keys -> to_out + residual in ONE row-local launch (skg_xattn_block_f16); the
            # text-dependent part needs both halves, so a shared front ends here
            if shared:
                ops.batch_copy(p1, M1, p1_full, M1, 1, M1)
                x, p1, shared = x_full, p1_full, False
            xargs = (HW, heads, self.ctx["L"], W[t + ".norm2.weight"], W[t + ".norm2.bias"], 1e-5, W[t + ".attn2.xpack"], cb["kvpack"],
                     W[t + ".attn2.to_out.0.bias"], scale)
            if keep:
                # guided step: the same launch also stores what the backward of the cond rows (second half) reads - norm2's
                # statistics, q, the attention output and its lse (skg_xattn_block_f16_keep), for those rows only
                p2, st2, q2_c, o2, lse2 = ops.xattn_block(p1, *xargs, keep_from=(rows // 2) * HW)
                q2 = q2_c
                xk_half = ("st2", "q2", "o2", "lse2")
            else:
                p2 = ops.xattn_block(p1, *xargs)
                st2 = q2 = q2_c = o2 = lse2 = None            # (only a stash would read them, and there is none)
        else:
            a2, st2 = ops.layernorm(p1, W[t + ".norm2.weight"], W[t + ".norm2.bias"], want_stats=True)
            q2_full = torch.empty(rows * HW, C, device=x.device, dtype=torch.float16) if shared else None
            q2 = ops.gemm(a2, W[t + ".attn2.to_q.weight"], q2_full[M1:] if shared else None)
            q2_c = q2
            if shared:      # the text-dependent part needs both halves: one copy each into the uncond half
                ops.batch_copy(p1, M1, p1_full, M1, 1, M1)
                ops.batch_copy(q2, M1, q2_full, M1, 1, M1)
                x, p1, q2 = x_full, p1_full, q2_full
            o2, lse2 = ops.attn_fwd(q2, cb["K"], cb["V"], rows, heads, HW, self.ctx["L"], self.ctx["Lp"], dh, scale,
                                    want_lse=True, v_rows=True)
            p2 = ops.gemm(o2, W[t + ".attn2.to_out.0.weight"], bias=W[t + ".attn2.to_out.0.bias"], residual=p1)
        ffb = _FF_BLOCK and (t + ".ff.pack") in W and (not keep or rows % 2 == 0)
        st3_half = False
        ffp = ffb and _FF_PROJ and (t + ".ff.packp") in W and (not keep or _FF_KEEP) and HW % 128 == 0
        opart = None
        if ffp:
            # ... and proj_out + the outer residual behind it in the same launch (skg_ff_block_proj_f16): the block output p3 never
            # reaches memory (no backward reads it); GroupNorm partial sums of the result when the consumer folds them
            gn = (HW, cfg.norm_groups) if want_part and self._gn_from_producer(rows, HW, C) else None
            if out is None:
                out = torch.empty(rows * HW, C, device=x.device, dtype=torch.float16)
            _, st3, f, opart = ops.ff_block_proj(p2, W[t + ".norm3.weight"], W[t + ".norm3.bias"], 1e-5, W[t + ".ff.packp"], W[t + ".ff.bias1"],
                                                 W[t + ".ff.net.2.bias"], W[p + ".proj_out.bias"], x, out=out, want_stats=keep,
                                                 keep_from=(rows // 2) * HW if keep else None, gn=gn)
        elif ffb:
            # C = 320: norm3 -> FF1 -> gate -> FF2 + residual in ONE row-local launch (skg_ff_block_f16: the row panel and the
            # output accumulators stay in registers, only weights stream).  In a guided step the same launch also stores what
            # the backward of the cond rows reads: the FF1 output (interleaved pack) and norm3's statistics
            M0 = (rows // 2) * HW
            p3 = torch.empty(rows * HW, C, device=x.device, dtype=torch.float16)
            ffargs = (W[t + ".norm3.weight"], W[t + ".norm3.bias"], 1e-5, W[t + ".ff.pack"], W[t + ".ff.bias1"], W[t + ".ff.net.2.bias"])
            f, st3 = None, None
            if not keep:
                ops.ff_block(p2, *ffargs, out=p3)
            elif _FF_KEEP:
                _, st3, f = ops.ff_block(p2, *ffargs, out=p3, want_stats=True, keep_from=M0)
            else:       # (A/B switch: the uncond half fused, the cond half on the launches that stash)
                ops.ff_block(p2[:M0], *ffargs, out=p3[:M0])
                a3, st3 = ops.layernorm(p2[M0:], W[t + ".norm3.weight"], W[t + ".norm3.bias"], want_stats=True)
                st3_half = True
                if _GEGLU_KEEP:
                    ggc, f = ops.gemm_geglu_keep(a3, W[t + ".ff.net.0.proj.weight"], W[t + ".ff.net.0.proj.bias"])
                else:
                    f = ops.gemm(a3, W[t + ".ff.net.0.proj.weight"], bias=W[t + ".ff.net.0.proj.bias"])
                    ggc = ops.geglu(f, interleaved=True)
                ops.gemm(ggc, W[t + ".ff.net.2.weight"], p3[M0:], bias=W[t + ".ff.net.2.bias"], residual=p2[M0:])
        else:
            a3, st3 = ops.layernorm(p2, W[t + ".norm3.weight"], W[t + ".norm3.bias"], want_stats=True)
        if ffb:
            pass
        elif not keep and C % 64 == 0:        # no backward will follow: gate inside the GEMM epilogue (half the bytes)
            f = None
            gg = ops.gemm(a3, W[t + ".ff.net.0.proj.weight"], bias=W[t + ".ff.net.0.proj.bias"], geglu=True)
        elif C % 64 == 0 and rows % 2 == 0:
            # guided step: only the cond half (second half of the rows) is differentiated, so only it needs the
            # pre-activation f; the uncond half takes the fused-GEGLU GEMM (a fifth of the bytes of GEMM + gate kernel)
            M0 = (rows // 2) * HW
            gg = torch.empty(rows * HW, 4 * C, device=x.device, dtype=torch.float16)
            ops.gemm(a3[:M0], W[t + ".ff.net.0.proj.weight"], bias=W[t + ".ff.net.0.proj.bias"], geglu=True, out=gg[:M0])
            # cond rows: the same fused epilogue also stores the pre-activation the gate's backward needs
            if _GEGLU_KEEP:
                _, f = ops.gemm_geglu_keep(a3[M0:], W[t + ".ff.net.0.proj.weight"], W[t + ".ff.net.0.proj.bias"], out=gg[M0:])
            else:
                f = ops.gemm(a3[M0:], W[t + ".ff.net.0.proj.weight"], bias=W[t + ".ff.net.0.proj.bias"])
                ops.geglu(f, out=gg[M0:], interleaved=True)
        else:
            f = ops.gemm(a3, W[t + ".ff.net.0.proj.weight"], bias=W[t + ".ff.net.0.proj.bias"])
            gg = ops.geglu(f, interleaved=True)
            f = f[(rows // 2) * HW:]
        if not ffb:
            p3 = ops.gemm(gg, W[t + ".ff.net.2.weight"], bias=W[t + ".ff.net.2.bias"], residual=p2)
        if ffp:
            pass
        elif want_part and self._gn_from_producer(rows, HW, C):
            out, opart = ops.gemm(p3, W[p + ".proj_out.weight"], out, bias=W[p + ".proj_out.bias"], residual=x,
                                  gn_stats=(HW, cfg.norm_groups))
        else:
            out = ops.gemm(p3, W[p + ".proj_out.weight"], out, bias=W[p + ".proj_out.bias"], residual=x)
        if keep:
            # half: the tensors of the text-independent part hold the cond rows only (r1 == rows // 2)
            # (`half` is derived from what each tensor actually holds - r1 rows or all of them - whatever the injector did, so that
            # a backward through an injected block could never slice an M1-row tensor a second time: ADVICE r3)
            st3h = ("st3",) if st3_half else ()      # fused FF block: norm3's statistics exist for the cond rows only
            ent = dict(x=x_c, gst=gst, pin=pin, st1=st1, qkv=qkv, o1=o1, lse1=lse1, p1=p1_c, st2=st2, q2=q2_c)
            half = self._stash_half(ent, r1, rows, HW)
            if r1 != rows and "x" not in half:      # the front diverged before p1 (an injector with differing halves): x is full size
                ent["x"] = x
            half = tuple(k for k in half if k not in xk_half) + xk_half      # (the stashing cross-attention launch: cond rows only)
            stash.tr[p] = dict(ent, o2=o2, lse2=lse2, p2=p2, st3=st3, f=f, H=H, heads=heads, half=half + st3h)
        return out, opart

    @staticmethod
    def _stash_half(ent: dict, r1: int, rows: int, HW: int) -> tuple:
        """Keys of a transformer stash entry whose tensors hold the cond rows only (r1 = rows // 2 of them), derived from what each
        tensor actually holds - never from how the forward believes it got there - so that the backward can not slice a cond-only
        tensor a second time.  Per-row tensors have r1 * HW rows; gst / lse1 are per image."""
        if r1 == rows:
            return ()
        return tuple(k for k, v in ent.items() if v is not None and v.shape[0] == (r1 if k in ("gst", "lse1") else r1 * HW))

    @staticmethod
    def _dup_partial(part):
        """GroupNorm partial sums of the cond rows -> of [uncond rows ; cond rows] (the halves are identical)."""
        if part is None:
            return None
        both = ops.GNPartial.__new__(ops.GNPartial)
        both.rows, both.nch, both.groups = 2 * part.rows, part.nch, part.groups
        both.buf = torch.cat([part.buf, part.buf])
        return both

    # ------------------------------------------------------------------ forward
    def forward(self, x32: torch.Tensor, t: int, rows: int, H: int, stash: Optional[Stash] = None,
                want_taps: bool = True, want_eps: bool = True, down_only: bool = False, shared_input: bool = False,
                on_taps: Optional[Callable] = None):
        """x32: fp16 [rows*H*H, 32] (latent channels zero-padded).  Returns (eps [rows*H*H, 8] or None,
        taps: list of 9 (tensor [rows*s*s, C], s)).

        shared_input: the caller guarantees that the two halves of x32 are IDENTICAL - the CFG-doubled batch of
        modules/pipeline.py:85, `torch.cat([latents] * 2)`.  The two halves of the evaluation then only differ from the
        first text-dependent operation on (the first cross-attention), so everything in front of it - conv_in, the first
        ResnetBlock, and GroupNorm / proj_in / self-attention / LayerNorm 2 / to_q of the first transformer block, all at
        the full 64 x 64 resolution - is evaluated ONCE on the cond rows and copied to the uncond rows.  Exact in exact
        arithmetic (every kernel's result for a row depends on that row only); bit-identical to the doubled evaluation
        when the half-size launches run the same kernel instantiations, else equal to fp16 rounding noise - another
        summation order of GroupNorm partial sums / K slices (tests/test_gpu_pipeline.py::test_shared_cfg_prefix_is_bit_identical).

        on_taps(taps): called as soon as the ninth tap exists (after the third up block), before the last up block and
        conv_out are launched - the guidance branch (LGP + backward-to-input) depends on nothing later, so the sampler can put
        it on a second stream while this one finishes eps (sampler.HipSampler.fork_guidance)."""
        cfg, W = self.cfg, self.W
        assert self.ctx is not None and self.ctx["rows"] == rows, "call prepare_context first"
        self.prepare_timesteps([t])
        if self.residual_fp32:
            assert not down_only, "accuracy mode: the whole UNet (with or without a stash, with or without an injector)"
            return self._forward_hp(x32, t, rows, H, want_taps, want_eps, stash, on_taps, shared_input)
        if stash is not None:
            stash.misc.update(rows=rows, H=H)
        tb = self.tbias[int(t)]
        boc = cfg.block_out_channels
        nb = len(boc)
        # torch.cat((h, skip)) of the up path without copies: the buffer [h | skip] of every up resnet exists from the
        # start; the down-path layer that produces a skip writes it straight into the right-hand columns (and keeps
        # using that strided view as its own output), the up-path layer that produces h into the left-hand ones.
        lpb1 = cfg.layers_per_block + 1
        rev = list(reversed(boc))
        n_skips = 1 + sum(cfg.layers_per_block + (1 if i < nb - 1 else 0) for i in range(nb))
        cats: List[Optional[torch.Tensor]] = [None] * (nb * lpb1)
        ch_h = [rev[0] if u == 0 else (rev[u // lpb1 - 1] if u % lpb1 == 0 else rev[u // lpb1]) for u in range(nb * lpb1)]

        def skip_slot(ch_s: int, size: int):
            """Right-hand columns of the concat buffer that will consume the skip produced next."""
            u = n_skips - 1 - len(skips)
            if down_only:
                return None
            cats[u] = torch.empty(rows * size * size, ch_h[u] + ch_s, device=self.dev, dtype=torch.float16)
            return cats[u][:, ch_h[u]:]

        skips: List[torch.Tensor] = []
        skip_parts: List[Optional[ops.GNPartial]] = []       # partial sums of each skip, when its producer left them
        G = cfg.norm_groups
        # hp: GroupNorm partial sums of h left behind by the kernel that produced it, whenever the next consumer of h is a
        # GroupNorm of the 64 x 64 / 32 x 32 levels (None otherwise: concatenated inputs, small maps)
        shared = shared_input and rows % 2 == 0 and rows >= 2 and not down_only and nb > 1 and cfg.layers_per_block >= 1
        cur = H
        if shared:
            S1, M1 = rows // 2, (rows // 2) * H * H
            slot = skip_slot(boc[0], H)                           # the skip of conv_in, all rows
            if self._gn_from_producer(S1, H * H, boc[0]):
                h1, hp1 = ops.conv3x3(x32[M1:], W["conv_in.weight"], S1, H, H, out=slot[M1:], bias=W["conv_in.bias"], gn_groups=G)
            else:
                h1, hp1 = ops.conv3x3(x32[M1:], W["conv_in.weight"], S1, H, H, out=slot[M1:], bias=W["conv_in.bias"]), None
            ops.batch_copy(h1, M1, slot, M1, 1, M1)
            skips.append(slot)
            skip_parts.append(self._dup_partial(hp1))
            x_full = torch.empty(rows * H * H, boc[0], device=self.dev, dtype=torch.float16)
            h1, hp1 = self._res_fwd("down_blocks.0.resnets.0", h1, S1, cur, tb, stash, out=x_full[M1:], xpart=hp1,
                                    want_part=True, half=True)
            ops.batch_copy(h1, M1, x_full, M1, 1, M1)
            h, hp = self._tr_fwd("down_blocks.0.attentions.0", h1, rows, cur, cfg.num_heads[0], stash,
                                 out=skip_slot(boc[0], cur), xpart=hp1, want_part=0 < cfg.layers_per_block - 1,
                                 shared=True, x_full=x_full)
            skips.append(h)
            skip_parts.append(hp)
        elif self._gn_from_producer(rows, H * H, boc[0]):
            h, hp = ops.conv3x3(x32, W["conv_in.weight"], rows, H, H, out=skip_slot(boc[0], H), bias=W["conv_in.bias"],
                                gn_groups=G)
        else:
            h, hp = ops.conv3x3(x32, W["conv_in.weight"], rows, H, H, out=skip_slot(boc[0], H), bias=W["conv_in.bias"]), None
        if not shared:
            skips.append(h)
            skip_parts.append(hp)
        taps_down = []
        for i in range(nb):
            for j in range(cfg.layers_per_block):
                if shared and i == 0 and j == 0:
                    continue                                      # done above, once for both halves
                if i < nb - 1:
                    h, hp = self._res_fwd(f"down_blocks.{i}.resnets.{j}", h, rows, cur, tb, stash, xpart=hp, want_part=True)
                    # the block's last transformer feeds the downsampling conv, the others the next resnet's norm1
                    h, hp = self._tr_fwd(f"down_blocks.{i}.attentions.{j}", h, rows, cur, cfg.num_heads[i], stash,
                                         out=skip_slot(boc[i], cur), xpart=hp, want_part=j < cfg.layers_per_block - 1)
                else:
                    h, hp = self._res_fwd(f"down_blocks.{i}.resnets.{j}", h, rows, cur, tb, stash,
                                          out=skip_slot(boc[i], cur), xpart=hp, want_part=True)
                skips.append(h)
                skip_parts.append(hp)
            if i < nb - 1:
                p = f"down_blocks.{i}.downsamplers.0.conv"
                half = cur // 2
                if self._gn_from_producer(rows, half * half, boc[i]):
                    h, hp = ops.conv3x3(h, W[p + ".weight"], rows, cur, cur, ops.CONV_S2, out=skip_slot(boc[i], half),
                                        bias=W[p + ".bias"], gn_groups=G)
                else:
                    h, hp = ops.conv3x3(h, W[p + ".weight"], rows, cur, cur, ops.CONV_S2, out=skip_slot(boc[i], half),
                                        bias=W[p + ".bias"]), None
                cur //= 2
                skips.append(h)
                skip_parts.append(hp)
            if i < 3:
                taps_down.append((h, cur))
        if down_only:
            # modules/sketch_encoder.py:39-98: the forward stops after the down path and returns the per-block
            # tuples of residual samples ((tensor [rows*s*s, C], s), ...) - what SatMixin.set_res_samples consumes
            out, k = [], 1
            for i in range(nb):
                n = cfg.layers_per_block + (1 if i < nb - 1 else 0)
                sizes = [H >> i] * cfg.layers_per_block + ([H >> (i + 1)] if i < nb - 1 else [])
                out.append(tuple((skips[k + j], sizes[j]) for j in range(n)))
                k += n
            return out
        h, hp = self._res_fwd("mid_block.resnets.0", h, rows, cur, tb, stash, xpart=hp, want_part=True)
        tap_r0 = (h, cur)
        h, hp = self._tr_fwd("mid_block.attentions.0", h, rows, cur, cfg.num_heads[-1], stash, xpart=hp, want_part=True)
        tap_at = (h, cur)
        h, _ = self._res_fwd("mid_block.resnets.1", h, rows, cur, tb, stash, out=cats[0][:, :ch_h[0]], xpart=hp)
        tap_r1 = (h, cur)
        hp = None
        taps_up = []
        rev_heads = tuple(reversed(cfg.num_heads))
        last_needed = 2 if not want_eps else nb - 1
        for i in range(nb):
            if i > last_needed:
                break
            for j in range(lpb1):
                u = i * lpb1 + j
                skips.pop()                                   # already sits in cats[u][:, ch_h[u]:]
                sp = skip_parts.pop() if skip_parts else None
                cat = cats[u]
                # norm1 of a resnet that reads [h | skip]: both halves' producers may have left their sums behind
                cpart = None
                if _GN_CONCAT and hp is not None and sp is not None and ops.gn_concat_ok(ch_h[u], cat.shape[1] - ch_h[u], G, hp.groups, sp.groups):
                    cpart = (hp, ch_h[u], sp)
                # where this layer's output goes: the next concat buffer of the same block, else a fresh tensor
                nxt = cats[u + 1][:, :ch_h[u + 1]] if j < lpb1 - 1 else None
                if i > 0:
                    # (norm1 reads the concatenation: regrouped producer sums where the widths allow, else its own pass)
                    h, hp = self._res_fwd(f"up_blocks.{i}.resnets.{j}", cat, rows, cur, tb, stash, xpart=cpart, want_part=True)
                    # the transformer's output goes to conv_norm_out (the very last one) or into the next concatenation,
                    # whose GroupNorm regroups its sums with the skip's; the block's last output feeds the upsampling conv
                    keep = (want_eps and i == nb - 1 and j == lpb1 - 1) or j < lpb1 - 1
                    h, hp = self._tr_fwd(f"up_blocks.{i}.attentions.{j}", h, rows, cur, rev_heads[i], stash, out=nxt,
                                         xpart=hp, want_part=keep)
                else:
                    h, hp = self._res_fwd(f"up_blocks.{i}.resnets.{j}", cat, rows, cur, tb, stash, out=nxt, xpart=cpart,
                                          want_part=j < lpb1 - 1)
            if i < nb - 1:
                p = f"up_blocks.{i}.upsamplers.0.conv"
                u = (i + 1) * lpb1
                # polyphase: 16 instead of 36 tap-products per low-res pixel (four launches; one grid on the small maps)
                if (p + ".weight:pp") in W and (UP2_SMALL_MAPS or (rows * cur * cur // 128) * (h.shape[1] // 160) >= 200):
                    h = ops.conv_up2(h, W[p + ".weight:pp"], rows, cur, cur, out=cats[u][:, :ch_h[u]], bias=W[p + ".bias"], W9=W[p + ".weight"])
                else:
                    h = ops.conv3x3(h, W[p + ".weight"], rows, cur, cur, ops.CONV_UP2, out=cats[u][:, :ch_h[u]],
                                    bias=W[p + ".bias"])
                hp = None              # (the next concatenation's widths do not line up with 32 groups per half)
                cur *= 2
            if i < 3:
                taps_up.append((h, cur))
                if i == 2 and on_taps is not None:
                    on_taps(taps_down + [tap_at, tap_r0, tap_r1] + taps_up)
        eps = None
        if want_eps:
            n, _ = ops.groupnorm(h, rows, cur * cur, cfg.norm_groups, 1e-5, W["conv_norm_out.weight"],
                                 W["conv_norm_out.bias"], True, partial=hp)
            eps = ops.conv3x3(n, W["conv_out.weight"], rows, cur, cur, bias=W["conv_out.bias"])
        taps = taps_down + [tap_at, tap_r0, tap_r1] + taps_up
        if stash is not None:
            stash.misc.update(rows=rows, H=H)
        return eps, (taps if want_taps else None)


    # ------------------------------------------------------------------ accuracy mode (residual_fp32)
    _Pair = ops.Pair      # (hi, lo) views with one pitch; `full` = the [M, 2C] tensor [hi | lo] when the two are its halves

    def _pair(self, M: int, C: int) -> ops.Pair:
        return ops.Pair.empty(M, C, self.dev)

    def _gn_hp(self, x, rows, HW, eps, name, silu, partial=None):
        """-> (normalised fp16 tensor, statistics [rows, groups, 2]); partial: the producer's epilogue sums (see ops.groupnorm_hilo).
        A norm listed in HP_NORM_PAIRS returns its output as a PAIR (ops.Pair with .full = the K-doubled operand [hi | lo])."""
        if name in self.hp_norm_pairs:
            n = self._pair(x.hi.shape[0], x.hi.shape[1])
            _, st = ops.groupnorm_hilo(x.hi, x.lo, rows, HW, self.cfg.norm_groups, eps, self.W[name + ".weight"],
                                       self.W[name + ".bias"], silu, out=n.hi, out_lo=n.lo, want_stats=True, partial=partial)
            return n, st
        return ops.groupnorm_hilo(x.hi, x.lo, rows, HW, self.cfg.norm_groups, eps, self.W[name + ".weight"],
                                  self.W[name + ".bias"], silu, want_stats=True, partial=partial)

    def _res_fwd_hp(self, p, x, rows, H, tb, out=None, full_of=None, stash=None, xpart=None, want_part=False, half=False):
        """As _res_fwd on pairs.  Returns (out pair, GroupNorm partial sums of out or None)."""
        W = self.W
        G = self.cfg.norm_groups
        HW, M = H * H, rows * H * H
        Cout = W[p + ".conv1.weight"].shape[0]
        n1, st1 = self._gn_hp(x, rows, HW, 1e-5, p + ".norm1", True, partial=xpart)
        h1 = self._pair(M, Cout)                                    # conv1 output feeds norm2: kept as a pair ("lin_n")
        fuse = self._gn_from_producer(rows, HW, Cout)
        part1 = None
        a1, w1 = (n1.full, W[p + ".conv1.weight:n2"]) if isinstance(n1, ops.Pair) else (n1, W[p + ".conv1.weight"])
        wino = _HP_WINO and not fuse      # (small maps only: the levels whose GroupNorm statistics do not come from the producers)
        if wino and not isinstance(n1, ops.Pair) and self._conv_wino(p + ".conv1.weight:wino", n1, rows, H, out=h1.hi, out_lo=h1.lo, bias=tb[p]) is not None:
            pass
        elif fuse:
            _, part1 = ops.conv3x3(a1, w1, rows, H, H, out=h1.hi, out_lo=h1.lo, bias=tb[p], gn_groups=G)
        else:
            ops.conv3x3(a1, w1, rows, H, H, out=h1.hi, out_lo=h1.lo, bias=tb[p])
        n2, st2 = self._gn_hp(h1, rows, HW, 1e-5, p + ".norm2", True, partial=part1)
        n2_pair = isinstance(n2, ops.Pair)
        w2n = W[p + ".conv2.weight:n2"] if n2_pair else None
        if n2_pair:
            n2 = n2.full
        if stash is not None:      # the backward differentiates the fp16 (hi) values, as in the default mode
            stash.res[p] = dict(x=x.hi, st1=st1, h1=h1.hi, st2=st2, H=H, half=half)
        out = out or self._pair(M, Cout)
        opart = None
        if wino and not n2_pair and self._wino_ok(p + ".conv2.weight:wino", rows, H):
            # Winograd conv2, pair output; its residual is the block input pair or the K-doubled shortcut GEMM's pair output
            if (p + ".conv_shortcut.weight") in W:
                sc = self._pair(M, Cout)
                xf = x.full if x.full is not None else full_of(x, x.hi.shape[1])
                ops.gemm(xf, W[p + ".conv_shortcut.weight:2"], out=sc.hi, out_lo=sc.lo, bias=W[p + ".conv_shortcut.bias"])
            else:
                sc = x
            if self._conv_wino(p + ".conv2.weight:wino", n2, rows, H, out=out.hi, out_lo=out.lo, bias=W[p + ".conv2.bias"],
                               residual=sc.hi, residual_lo=sc.lo) is not None:
                return out, None
        if (p + ".conv2.weight:sc2") in W:
            # conv2 + the K-doubled shortcut [x_hi | x_lo] . [W_sc | W_sc] in one launch, pair output
            xf = x.full if x.full is not None else full_of(x, x.hi.shape[1])
            wsc2 = w2n if n2_pair else W[p + ".conv2.weight:sc2"]
            try:
                if want_part and fuse:
                    _, opart = ops.conv3x3_sc(n2, xf, wsc2, rows, H, H, out=out.hi, out_lo=out.lo,
                                              bias=W[p + ".conv2.bias:sc"], gn_groups=G)
                else:
                    ops.conv3x3_sc(n2, xf, wsc2, rows, H, H, out=out.hi, out_lo=out.lo, bias=W[p + ".conv2.bias:sc"])
                return out, opart
            except ops.SkgError as e:
                if e.rc != -2:
                    raise
            if n2_pair:      # the two-launch fall-back takes conv2's own [W | W] taps (the leading columns of the folded pack)
                w2n = w2n[:, : 9 * n2.shape[1]]
        if (p + ".conv_shortcut.weight") in W:
            sc = self._pair(M, Cout)                                 # the stream as a matmul operand: [hi | lo] . [W | W]
            xf = x.full if x.full is not None else full_of(x, x.hi.shape[1])
            ops.gemm(xf, W[p + ".conv_shortcut.weight:2"], out=sc.hi, out_lo=sc.lo, bias=W[p + ".conv_shortcut.bias"])
        else:
            sc = x
        w2 = w2n if n2_pair else W[p + ".conv2.weight"]
        if not w2.is_contiguous():              # (a column slice of the folded pack: the kernel wants the 9-tap pack contiguous)
            w2 = w2.contiguous()
        if want_part and fuse:
            _, opart = ops.conv3x3(n2, w2, rows, H, H, out=out.hi, out_lo=out.lo, bias=W[p + ".conv2.bias"],
                                   residual=sc.hi, residual_lo=sc.lo, gn_groups=G)
        else:
            ops.conv3x3(n2, w2, rows, H, H, out=out.hi, out_lo=out.lo, bias=W[p + ".conv2.bias"],
                        residual=sc.hi, residual_lo=sc.lo)
        return out, opart

    def _tr_fwd_hp(self, p, x, rows, H, heads, out=None, stash=None, xpart=None, want_part=False, shared=False, x_full=None):
        """As _tr_fwd on pairs (shared / x_full: the shared CFG front, see _tr_fwd: x holds the cond rows only)."""
        W = self.W
        HW, M = H * H, rows * H * H
        C = x.hi.shape[1]
        dh = C // heads
        scale = dh ** -0.5
        t = p + ".transformer_blocks.0"
        keep = stash is not None
        r1 = rows // 2 if shared else rows
        M1 = r1 * HW
        g, gst = self._gn_hp(x, r1, HW, 1e-6, p + ".norm", False, partial=xpart)
        pin = self._pair(M1, C)
        if isinstance(g, ops.Pair):
            ops.gemm(g.full, W[p + ".proj_in.weight:n2"], out=pin.hi, out_lo=pin.lo, bias=W[p + ".proj_in.bias"])
        else:
            ops.gemm(g, W[p + ".proj_in.weight"], out=pin.hi, out_lo=pin.lo, bias=W[p + ".proj_in.bias"])
        a1, st1 = ops.layernorm_hilo(pin.hi, pin.lo, W[t + ".norm1.weight"], W[t + ".norm1.bias"], want_stats=True)
        qkv = ops.gemm(a1, W[t + ".attn1.qkv"])
        o1, lse1 = ops.attn_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], r1, heads, HW, HW, HW, dh, scale,
                                want_lse=True, v_rows=True)
        P = HipUNet._Pair
        p1_full = self._pair(M, C) if shared else None
        p1 = P(p1_full.hi[M1:], p1_full.lo[M1:], p1_full.full[M1:]) if shared else self._pair(M, C)
        ops.gemm(o1, W[t + ".attn1.to_out.0.weight"], out=p1.hi, out_lo=p1.lo, bias=W[t + ".attn1.to_out.0.bias"],
                 residual=pin.hi, residual_lo=pin.lo)
        x_c, p1_c = x, p1
        if self.inject is not None:
            if shared and getattr(self.inject, "halves_equal", False):
                nf = self._pair(M, C)
                p1 = self.inject(t, p1, rows, HW, heads, cond_only=True, out=P(nf.hi[M1:], nf.lo[M1:], nf.full[M1:]))
                p1_full = nf
            else:
                if shared:                                    # the injected K / V differ between the halves: diverge here
                    ops.batch_copy(p1.full, M1, p1_full.full, M1, 1, M1)
                    x, p1, shared = x_full, p1_full, False
                p1 = self.inject(t, p1, rows, HW, heads)
            p1_c = p1
        cb = self.ctx["blocks"][t + ".attn2"]
        xab = (_XATTN_BLOCK and (not keep or (_XATTN_KEEP_HP and rows % 2 == 0)) and "kvpack" in cb and heads in _XATTN_HEADS and HW % 128 == 0)
        xk_half = ()
        if xab:
            # norm2 -> to_q -> text attention -> to_out + residual in ONE row-local launch on the pair; in a guided step the same
            # launch stores what the backward of the cond rows reads (skg_xattn_block_f16_hilo_keep), as _tr_fwd does
            if shared:
                ops.batch_copy(p1.full, M1, p1_full.full, M1, 1, M1)
                x, p1, shared = x_full, p1_full, False
            xargs = (HW, heads, self.ctx["L"], W[t + ".norm2.weight"], W[t + ".norm2.bias"], 1e-5, W[t + ".attn2.xpack"], cb["kvpack"],
                     W[t + ".attn2.to_out.0.bias"], scale)
            if keep:
                p2, st2, q2_c, o2, lse2 = ops.xattn_block(p1, *xargs, keep_from=(rows // 2) * HW)
                q2 = q2_c
                xk_half = ("st2", "q2", "o2", "lse2")
            else:
                p2 = ops.xattn_block(p1, *xargs)
                st2 = q2 = q2_c = o2 = lse2 = None
        else:
            a2, st2 = ops.layernorm_hilo(p1.hi, p1.lo, W[t + ".norm2.weight"], W[t + ".norm2.bias"], want_stats=True)
            q2_full = torch.empty(M, C, device=self.dev, dtype=torch.float16) if shared else None
            q2 = ops.gemm(a2, W[t + ".attn2.to_q.weight"], q2_full[M1:] if shared else None)
            q2_c = q2
            if shared:      # the text-dependent part needs both halves: one copy each into the uncond half
                ops.batch_copy(p1.full, M1, p1_full.full, M1, 1, M1)
                ops.batch_copy(q2, M1, q2_full, M1, 1, M1)
                x, p1, q2 = x_full, p1_full, q2_full
            o2, lse2 = ops.attn_fwd(q2, cb["K"], cb["V"], rows, heads, HW, self.ctx["L"], self.ctx["Lp"], dh, scale,
                                    want_lse=True, v_rows=True)
            p2 = self._pair(M, C)
            ops.gemm(o2, W[t + ".attn2.to_out.0.weight"], out=p2.hi, out_lo=p2.lo, bias=W[t + ".attn2.to_out.0.bias"],
                     residual=p1.hi, residual_lo=p1.lo)
        ffb = _FF_BLOCK and (t + ".ff.pack") in W and (not keep or (rows % 2 == 0 and _FF_KEEP))
        f = None
        ffp = ffb and _FF_PROJ and _FF_PROJ_HP and (t + ".ff.packp") in W and HW % 128 == 0
        if ffp:
            # ... with proj_out (on the pair p3 = hi + lo: W . hi + W . lo on the same fragments) and the outer residual pair in the
            # same launch (skg_ff_block_proj_f16_hilo): the K-doubled proj_out GEMM and the [M, 2C] round trip of p3 disappear
            out = out or self._pair(M, C)
            gn = (HW, self.cfg.norm_groups) if want_part and self._gn_from_producer(rows, HW, C) else None
            _, st3, f, opart = ops.ff_block_proj(p2, W[t + ".norm3.weight"], W[t + ".norm3.bias"], 1e-5, W[t + ".ff.packp"], W[t + ".ff.bias1"],
                                                 W[t + ".ff.net.2.bias"], W[p + ".proj_out.bias"], x, out=out, want_stats=keep,
                                                 keep_from=(rows // 2) * HW if keep else None, gn=gn)
            if keep:
                own = r1 != rows and self.inject is None
                ent = dict(x=(x_c if own else x).hi, gst=gst, pin=pin.hi, st1=st1, qkv=qkv, o1=o1, lse1=lse1, p1=(p1_c if own else p1).hi,
                           st2=st2, q2=q2_c if own else q2)
                half = self._stash_half(ent, r1, rows, HW)
                half = tuple(k for k in half if k not in xk_half) + xk_half
                stash.tr[p] = dict(ent, o2=o2, lse2=lse2, p2=p2.hi, st3=st3, f=f, H=H, heads=heads, half=half)
            return out, opart
        if ffb:
            # C = 320: norm3 -> FF1 -> gate -> FF2 + residual in ONE row-local launch on the pair (guided steps: the same launch
            # stores the cond rows' FF1 output and norm3's statistics for the backward)
            ffargs = (W[t + ".norm3.weight"], W[t + ".norm3.bias"], 1e-5, W[t + ".ff.pack"], W[t + ".ff.bias1"], W[t + ".ff.net.2.bias"])
            st3 = None
            if keep:
                p3, st3, f = ops.ff_block(p2, *ffargs, want_stats=True, keep_from=(rows // 2) * HW)
            else:
                p3 = ops.ff_block(p2, *ffargs)
        else:
            a3, st3 = ops.layernorm_hilo(p2.hi, p2.lo, W[t + ".norm3.weight"], W[t + ".norm3.bias"], want_stats=True)
        if ffb:
            pass
        elif keep and C % 64 == 0 and rows % 2 == 0:     # guided step: the cond half's gate also keeps its pre-activation
            M0 = (rows // 2) * HW
            gg = torch.empty(M, 4 * C, device=self.dev, dtype=torch.float16)
            ops.gemm(a3[:M0], W[t + ".ff.net.0.proj.weight"], bias=W[t + ".ff.net.0.proj.bias"], geglu=True, out=gg[:M0])
            _, f = ops.gemm_geglu_keep(a3[M0:], W[t + ".ff.net.0.proj.weight"], W[t + ".ff.net.0.proj.bias"], out=gg[M0:])
        elif C % 64 == 0:
            gg = ops.gemm(a3, W[t + ".ff.net.0.proj.weight"], bias=W[t + ".ff.net.0.proj.bias"], geglu=True)
        else:
            ff = ops.gemm(a3, W[t + ".ff.net.0.proj.weight"], bias=W[t + ".ff.net.0.proj.bias"])
            gg = ops.geglu(ff, interleaved=True)
            f = ff[(rows // 2) * HW:]
        if keep:
            # half: which tensors of the text-independent part hold the cond rows only - derived from their shapes, as _tr_fwd does
            # (with an injector the shared front ends earlier: x / p1 / q2 are full size while gst .. lse1 are not)
            own = r1 != rows and self.inject is None
            ent = dict(x=(x_c if own else x).hi, gst=gst, pin=pin.hi, st1=st1, qkv=qkv, o1=o1, lse1=lse1, p1=(p1_c if own else p1).hi,
                       st2=st2, q2=q2_c if own else q2)
            half = self._stash_half(ent, r1, rows, HW)
            half = tuple(k for k in half if k not in xk_half) + xk_half
            stash.tr[p] = dict(ent, o2=o2, lse2=lse2, p2=p2.hi, st3=st3, f=f, H=H, heads=heads, half=half)
        if not ffb:
            p3 = self._pair(M, C)
            ops.gemm(gg, W[t + ".ff.net.2.weight"], out=p3.hi, out_lo=p3.lo, bias=W[t + ".ff.net.2.bias"],
                     residual=p2.hi, residual_lo=p2.lo)
        out = out or self._pair(M, C)
        opart = None
        if want_part and self._gn_from_producer(rows, HW, C):
            _, opart = ops.gemm(p3.full, W[p + ".proj_out.weight:2"], out=out.hi, out_lo=out.lo, bias=W[p + ".proj_out.bias"],
                                residual=x.hi, residual_lo=x.lo, gn_stats=(HW, self.cfg.norm_groups))
        else:
            ops.gemm(p3.full, W[p + ".proj_out.weight:2"], out=out.hi, out_lo=out.lo, bias=W[p + ".proj_out.bias"],
                     residual=x.hi, residual_lo=x.lo)
        return out, opart

    def _forward_hp(self, x32, t, rows, H, want_taps, want_eps, stash=None, on_taps=None, shared_input=False):
        """The forward of forward() with the residual stream as (hi, lo) pairs; the same graph, the same kernels for every
        contraction, pair-aware epilogues / norms (skg_*_hilo), the same GroupNorm statistics from the producers' epilogues and
        the same shared CFG front.  Concatenations [h | skip] are pair buffers [h_hi | skip_hi | h_lo | skip_lo], filled in
        place by their producers.  Round 6: the blocks of the `hp_plain_levels` deepest resolution levels run the DEFAULT kernels
        on plain fp16 tensors (their pairs buy ~4 % of what the mode's pairs buy, tools/eps_decompose_stream.py): the stride-2
        convolution that enters the zone reads the pair and writes fp16, the upsampler that leaves it reads fp16 and writes a pair;
        skips produced inside the zone are consumed inside it (the U-Net's skips connect equal resolutions)."""
        cfg, W = self.cfg, self.W
        if stash is not None:
            stash.misc.update(rows=rows, H=H)
        tb = self.tbias[int(t)]
        boc = cfg.block_out_channels
        nb = len(boc)
        G = cfg.norm_groups
        lpb1 = cfg.layers_per_block + 1
        rev = list(reversed(boc))
        n_skips = 1 + sum(cfg.layers_per_block + (1 if i < nb - 1 else 0) for i in range(nb))
        ch_h = [rev[0] if u == 0 else (rev[u // lpb1 - 1] if u % lpb1 == 0 else rev[u // lpb1]) for u in range(nb * lpb1)]
        cats: List[Optional[torch.Tensor]] = [None] * (nb * lpb1)
        P = HipUNet._Pair
        n_made = [0]
        skip_parts: List[Optional[ops.GNPartial]] = []       # partial sums of each skip, when its producer left them
        pl = self.hp_plain_levels
        plain_up = lambda i: i <= pl - 1                      # up block i (and concat buffer u with u // lpb1 == i) is in the plain zone
        plain_down = lambda i: i >= nb - pl                   # down block i

        def skip_slot(ch_s: int, size: int):
            """The view inside the concat buffer that will consume the skip produced next: a pair view, or - the consumer is a
            block of the plain zone - the fp16 right-hand columns."""
            u = n_skips - 1 - n_made[0]
            n_made[0] += 1
            ct = ch_h[u] + ch_s
            if plain_up(u // lpb1):
                cats[u] = torch.empty(rows * size * size, ct, device=self.dev, dtype=torch.float16)
                return cats[u][:, ch_h[u]:]
            cats[u] = torch.empty(rows * size * size, 2 * ct, device=self.dev, dtype=torch.float16)
            return P(cats[u][:, ch_h[u]:ct], cats[u][:, ct + ch_h[u]:], None)

        def full_of(pv, C):
            """[hi | lo] as ONE operand: the pair's own buffer, or a packed copy when it lives inside a concat buffer."""
            if pv.full is not None:
                return pv.full
            buf = torch.empty(pv.hi.shape[0], 2 * C, device=self.dev, dtype=torch.float16)
            ops.axpby(pv.hi, None, out=buf[:, :C])
            ops.axpby(pv.lo, None, out=buf[:, C:])
            return buf

        def rows_of(pv, m0, m1):
            return P(pv.hi[m0:m1], pv.lo[m0:m1], None if pv.full is None else pv.full[m0:m1])

        def conv_pair(x, wkey, bkey, size, mode, o, r=rows, want=False):
            """conv with a pair output (+ the GroupNorm sums of its hi part where the level takes them from the producers)"""
            osz = size // 2 if mode == ops.CONV_S2 else size
            if want and self._gn_from_producer(r, osz * osz, o.hi.shape[1]):
                return ops.conv3x3(x, W[wkey], r, size, size, mode, out=o.hi, out_lo=o.lo, bias=W[bkey], gn_groups=G)[1]
            ops.conv3x3(x, W[wkey], r, size, size, mode, out=o.hi, out_lo=o.lo, bias=W[bkey])
            return None

        def conv_plain(x, wkey, bkey, size, mode, o):
            """conv with an fp16 output (the plain zone; forward()'s form)"""
            osz = size // 2 if mode == ops.CONV_S2 else size
            if self._gn_from_producer(rows, osz * osz, o.shape[1]):
                return ops.conv3x3(x, W[wkey], rows, size, size, mode, out=o, bias=W[bkey], gn_groups=G)[1]
            ops.conv3x3(x, W[wkey], rows, size, size, mode, out=o, bias=W[bkey])
            return None

        hi_of = lambda v: v.hi if isinstance(v, ops.Pair) else v
        shared = shared_input and rows % 2 == 0 and rows >= 2 and nb > 1 and cfg.layers_per_block >= 1
        cur = H
        taps_down = []
        h = skip_slot(boc[0], H)
        if shared:
            S1, M1 = rows // 2, (rows // 2) * H * H
            hc = rows_of(h, M1, 2 * M1)
            hp1 = conv_pair(x32[M1:], "conv_in.weight", "conv_in.bias", H, ops.CONV_S1, hc, r=S1, want=True)
            ops.batch_copy(hc.hi, M1, h.hi, M1, 1, M1)
            ops.batch_copy(hc.lo, M1, h.lo, M1, 1, M1)
            skip_parts.append(self._dup_partial(hp1))
            x_full = self._pair(rows * H * H, boc[0])
            h1, hp1 = self._res_fwd_hp("down_blocks.0.resnets.0", hc, S1, cur, tb, out=rows_of(x_full, M1, 2 * M1), full_of=full_of,
                                       stash=stash, xpart=hp1, want_part=True, half=True)
            ops.batch_copy(h1.full, M1, x_full.full, M1, 1, M1)
            h, hp = self._tr_fwd_hp("down_blocks.0.attentions.0", h1, rows, cur, cfg.num_heads[0], out=skip_slot(boc[0], cur),
                                    stash=stash, xpart=hp1, want_part=0 < cfg.layers_per_block - 1, shared=True, x_full=x_full)
            skip_parts.append(hp)
        else:
            hp = conv_pair(x32, "conv_in.weight", "conv_in.bias", H, ops.CONV_S1, h, want=True)
            skip_parts.append(hp)
        for i in range(nb):
            for j in range(cfg.layers_per_block):
                if shared and i == 0 and j == 0:
                    continue
                rp, ap = f"down_blocks.{i}.resnets.{j}", f"down_blocks.{i}.attentions.{j}"
                if plain_down(i):      # forward()'s form on fp16 tensors
                    if i < nb - 1:
                        h, hp = self._res_fwd(rp, h, rows, cur, tb, stash, xpart=hp, want_part=True)
                        h, hp = self._tr_fwd(ap, h, rows, cur, cfg.num_heads[i], stash, out=skip_slot(boc[i], cur), xpart=hp,
                                             want_part=j < cfg.layers_per_block - 1)
                    else:
                        h, hp = self._res_fwd(rp, h, rows, cur, tb, stash, out=skip_slot(boc[i], cur), xpart=hp, want_part=True)
                elif i < nb - 1:
                    h, hp = self._res_fwd_hp(rp, h, rows, cur, tb, full_of=full_of, stash=stash, xpart=hp, want_part=True)
                    h, hp = self._tr_fwd_hp(ap, h, rows, cur, cfg.num_heads[i], out=skip_slot(boc[i], cur), stash=stash, xpart=hp,
                                            want_part=j < cfg.layers_per_block - 1)
                else:
                    h, hp = self._res_fwd_hp(rp, h, rows, cur, tb, out=skip_slot(boc[i], cur), full_of=full_of, stash=stash, xpart=hp,
                                             want_part=True)
                skip_parts.append(hp)
            if i < nb - 1:
                p = f"down_blocks.{i}.downsamplers.0.conv"
                o = skip_slot(boc[i], cur // 2)               # (a pair, or fp16 when the consumer - the same level of the up path - is plain)
                if plain_down(i):
                    hp = conv_plain(h, p + ".weight", p + ".bias", cur, ops.CONV_S2, o)
                elif isinstance(o, ops.Pair):
                    hp = conv_pair(full_of(h, boc[i]), p + ".weight:2", p + ".bias", cur, ops.CONV_S2, o, want=True)
                else:                                         # entering the plain zone: the pair as K-doubled operand, fp16 out
                    hp = conv_plain(full_of(h, boc[i]), p + ".weight:2", p + ".bias", cur, ops.CONV_S2, o)
                h = o
                cur //= 2
                skip_parts.append(hp)
            if i < 3:
                taps_down.append((hi_of(h), cur))
        if pl > 0:
            h, hp = self._res_fwd("mid_block.resnets.0", h, rows, cur, tb, stash, xpart=hp, want_part=True)
            tap_r0 = (h, cur)
            h, hp = self._tr_fwd("mid_block.attentions.0", h, rows, cur, cfg.num_heads[-1], stash, xpart=hp, want_part=True)
            tap_at = (h, cur)
            h, _ = self._res_fwd("mid_block.resnets.1", h, rows, cur, tb, stash, out=cats[0][:, :ch_h[0]], xpart=hp)
            tap_r1 = (h, cur)
        else:
            h, hp = self._res_fwd_hp("mid_block.resnets.0", h, rows, cur, tb, full_of=full_of, stash=stash, xpart=hp, want_part=True)
            tap_r0 = (h.hi, cur)
            h, hp = self._tr_fwd_hp("mid_block.attentions.0", h, rows, cur, cfg.num_heads[-1], stash=stash, xpart=hp, want_part=True)
            tap_at = (h.hi, cur)
            ct0 = cats[0].shape[1] // 2
            h, _ = self._res_fwd_hp("mid_block.resnets.1", h, rows, cur, tb, out=P(cats[0][:, :ch_h[0]], cats[0][:, ct0:ct0 + ch_h[0]]),
                                    stash=stash, xpart=hp)
            tap_r1 = (h.hi, cur)
        hp = None
        taps_up = []
        rev_heads = tuple(reversed(cfg.num_heads))
        last_needed = 2 if not want_eps else nb - 1
        for i in range(nb):
            if i > last_needed:
                break
            for j in range(lpb1):
                u = i * lpb1 + j
                sp = skip_parts.pop() if skip_parts else None
                rp, ap = f"up_blocks.{i}.resnets.{j}", f"up_blocks.{i}.attentions.{j}"
                if plain_up(i):        # forward()'s form
                    cat = cats[u]
                    cpart = None
                    if _GN_CONCAT and hp is not None and sp is not None and ops.gn_concat_ok(ch_h[u], cat.shape[1] - ch_h[u], G, hp.groups, sp.groups):
                        cpart = (hp, ch_h[u], sp)
                    nxt = cats[u + 1][:, :ch_h[u + 1]] if j < lpb1 - 1 else None
                    if i > 0:
                        h, hp = self._res_fwd(rp, cat, rows, cur, tb, stash, xpart=cpart, want_part=True)
                        h, hp = self._tr_fwd(ap, h, rows, cur, rev_heads[i], stash, out=nxt, xpart=hp, want_part=j < lpb1 - 1)
                    else:
                        h, hp = self._res_fwd(rp, cat, rows, cur, tb, stash, out=nxt, xpart=cpart, want_part=j < lpb1 - 1)
                    continue
                ct = cats[u].shape[1] // 2
                cat = P(cats[u][:, :ct], cats[u][:, ct:], cats[u])
                cpart = None
                if _GN_CONCAT and hp is not None and sp is not None and ops.gn_concat_ok(ch_h[u], ct - ch_h[u], G, hp.groups, sp.groups):
                    cpart = (hp, ch_h[u], sp)
                nxt = None
                if j < lpb1 - 1:
                    ctn = cats[u + 1].shape[1] // 2
                    nxt = P(cats[u + 1][:, :ch_h[u + 1]], cats[u + 1][:, ctn:ctn + ch_h[u + 1]])
                if i > 0:
                    h, hp = self._res_fwd_hp(rp, cat, rows, cur, tb, stash=stash, xpart=cpart, want_part=True)
                    keepp = (want_eps and i == nb - 1 and j == lpb1 - 1) or j < lpb1 - 1
                    h, hp = self._tr_fwd_hp(ap, h, rows, cur, rev_heads[i], out=nxt, stash=stash, xpart=hp, want_part=keepp)
                else:
                    h, hp = self._res_fwd_hp(rp, cat, rows, cur, tb, out=nxt, stash=stash, xpart=cpart, want_part=j < lpb1 - 1)
            if i < nb - 1:
                p = f"up_blocks.{i}.upsamplers.0.conv"
                u = (i + 1) * lpb1
                if plain_up(i + 1):    # inside the plain zone: forward()'s form
                    o = cats[u][:, :ch_h[u]]
                    if (p + ".weight:pp") in W and (UP2_SMALL_MAPS or (rows * cur * cur // 128) * (h.shape[1] // 160) >= 200):
                        ops.conv_up2(h, W[p + ".weight:pp"], rows, cur, cur, out=o, bias=W[p + ".bias"], W9=W[p + ".weight"])
                    else:
                        ops.conv3x3(h, W[p + ".weight"], rows, cur, cur, ops.CONV_UP2, out=o, bias=W[p + ".bias"])
                else:
                    ctn = cats[u].shape[1] // 2
                    o = P(cats[u][:, :ch_h[u]], cats[u][:, ctn:ctn + ch_h[u]])
                    hin = hi_of(h)                            # (leaving the plain zone: h is fp16 and the hi-only launch is the natural one)
                    if (plain_up(i) or not HP_UP_TRIPLE) and (p + ".weight:pp") in W and (p + ".weight:pp3") not in W:
                        try:
                            ops.conv_up2_pairout(hin, W[p + ".weight:pp"], rows, cur, cur, o, bias=W[p + ".bias"])
                        except ops.SkgError as e:      # declined (an operand >= 2 GiB): the 9-tap gather form on the SAME hi-only operand
                            if e.rc != -2:               # (both routes of the mode must see the same operand; pair output either way)
                                raise
                            ops.conv3x3(hin, W[p + ".weight"], rows, cur, cur, ops.CONV_UP2, out=o.hi, out_lo=o.lo, bias=W[p + ".bias"])
                    elif plain_up(i):
                        ops.conv3x3(hin, W[p + ".weight"], rows, cur, cur, ops.CONV_UP2, out=o.hi, out_lo=o.lo, bias=W[p + ".bias"])
                    elif (p + ".weight:pp3") in W:      # polyphase, K axis [x_hi | x_lo | x_hi] . [W_hi | W_hi | W_lo]: 12 instead of 18 tap-products
                        ops.conv_up2_hilo(full_of(h, h.hi.shape[1]), W[p + ".weight:pp3"], rows, cur, cur, o, bias=W[p + ".bias"],
                                          W9x2=lambda p=p: self._w9x2(p + ".weight"))
                    else:
                        ops.conv3x3(full_of(h, h.hi.shape[1]), W[p + ".weight:2"], rows, cur, cur, ops.CONV_UP2, out=o.hi, out_lo=o.lo,
                                    bias=W[p + ".bias"])
                h = o
                hp = None
                cur *= 2
            if i < 3:
                taps_up.append((hi_of(h), cur))
                if i == 2 and on_taps is not None:
                    on_taps(taps_down + [tap_at, tap_r0, tap_r1] + taps_up)
        eps = None
        if want_eps:
            # conv_norm_out -> SiLU -> conv_out: both the normalised activation (the K-doubled operand [hi | lo] . [W | W]) and eps
            # itself (a (hi, lo) pair in one [M, 2 * 8] buffer, combined in fp32 by the CFG / scheduler kernel) keep fp32 accuracy -
            # their fp16 roundings would reach eps one to one (2.8e-4 relative each)
            n = self._pair(rows * cur * cur, boc[0])
            ops.groupnorm_hilo(h.hi, h.lo, rows, cur * cur, G, 1e-5, W["conv_norm_out.weight"], W["conv_norm_out.bias"], True,
                               out=n.hi, out_lo=n.lo, partial=hp)
            eps = self._pair(rows * cur * cur, COUT_PAD)
            ops.conv3x3(n.full, W["conv_out.weight:2"], rows, cur, cur, out=eps.hi, out_lo=eps.lo, bias=W["conv_out.bias"])
        taps = taps_down + [tap_at, tap_r0, tap_r1] + taps_up
        if stash is not None:
            stash.misc.update(rows=rows, H=H)
        return eps, (taps if want_taps else None)

    # ------------------------------------------------------------------ modules, backward (cond rows)
    def _res_bwd(self, p, dout, S, st: dict):
        cfg, W = self.cfg, self.W
        H = st["H"]
        HW, G = H * H, cfg.norm_groups
        if st.get("half"):        # the shared CFG prefix was evaluated on the cond rows only
            x, h1, st1, st2 = st["x"], st["h1"], st["st1"], st["st2"]
        else:
            x, h1 = st["x"][S * HW:], st["h1"][S * HW:]
            st1, st2 = st["st1"][S:], st["st2"][S:]
        dn2 = self._conv_wino(p + ".conv2.weight:winoT", dout, S, H)
        if dn2 is None:
            dn2 = ops.conv3x3(dout, W[p + ".conv2.weight:T"], S, H, H)
        dh1 = ops.groupnorm_bwd(h1, dn2, S, HW, G, st2, W[p + ".norm2.weight"], W[p + ".norm2.bias"], True)
        dn1 = self._conv_wino(p + ".conv1.weight:winoT", dh1, S, H)
        if dn1 is None:
            dn1 = ops.conv3x3(dh1, W[p + ".conv1.weight:T"], S, H, H)
        if (p + ".conv_shortcut.weight") in W:
            sc = ops.gemm(dout, W[p + ".conv_shortcut.weight:T"])
        else:
            sc = dout
        return ops.groupnorm_bwd(x, dn1, S, HW, G, st1, W[p + ".norm1.weight"], W[p + ".norm1.bias"], True,
                                 residual=sc)

    def _tr_bwd(self, p, dout, S, st: dict):
        cfg, W = self.cfg, self.W
        H, heads = st["H"], st["heads"]
        HW = H * H
        M0 = S * HW
        C = dout.shape[1]
        dh = C // heads
        scale = dh ** -0.5
        t = p + ".transformer_blocks.0"
        half = st.get("half", ())
        c = lambda a: a[M0:]
        cc = lambda k: st[k] if k in half else st[k][M0:]        # activations [rows*HW, .] of the cond rows
        cs = lambda k: st[k] if k in half else st[k][S:]         # per-row statistics / lse of the cond rows
        dp3 = ops.gemm(dout, W[p + ".proj_out.weight:T"])
        dgg = ops.gemm(dp3, W[t + ".ff.net.2.weight:T"])
        df = ops.geglu_bwd(st["f"], dgg, interleaved=True)            # f is stashed for the cond rows only
        da3 = ops.gemm(df, W[t + ".ff.net.0.proj.weight:T"])
        dp2 = ops.layernorm_bwd(c(st["p2"]), da3, W[t + ".norm3.weight"], cc("st3"), residual=dp3)
        # cross-attention: only dQ (K/V come from the constant text embeddings)
        do2 = ops.gemm(dp2, W[t + ".attn2.to_out.0.weight:T"])
        cb = self.ctx["blocks"][t + ".attn2"]
        L, Lp = self.ctx["L"], self.ctx["Lp"]
        if _ATTN_DQ_DELTA:      # delta = sum_d dO O in the prologue of the dQ launch (one launch and one read of dO fewer)
            dq2, _ = ops.attn_bwd_dq_delta(cc("q2"), cb["K"][S * Lp:], cb["V"][S * Lp:], do2, cc("o2"), cs("lse2"), S, heads, HW, L, Lp, dh, scale)
        else:
            delta2 = ops.attn_bwd_delta(cc("o2"), do2, S, heads, HW, dh)
            dq2 = ops.attn_bwd_dq(cc("q2"), cb["K"][S * Lp:], cb["V"][S * Lp:], do2, cs("lse2"),
                                  delta2, S, heads, HW, L, Lp, dh, scale)
        da2 = ops.gemm(dq2, W[t + ".attn2.to_q.weight:T"])
        dp1 = ops.layernorm_bwd(cc("p1"), da2, W[t + ".norm2.weight"], cc("st2"), residual=dp2)
        # self-attention
        do1 = ops.gemm(dp1, W[t + ".attn1.to_out.0.weight:T"])
        qkv = cc("qkv")
        Q, K, V = qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:]
        lse1 = cs("lse1")
        dqkv = torch.empty(M0, 3 * C, device=self.dev, dtype=torch.float16)
        if _ATTN_DQ_DELTA:
            _, delta1 = ops.attn_bwd_dq_delta(Q, K, V, do1, cc("o1"), lse1, S, heads, HW, HW, HW, dh, scale, out=dqkv[:, :C])
        else:
            delta1 = ops.attn_bwd_delta(cc("o1"), do1, S, heads, HW, dh)
            ops.attn_bwd_dq(Q, K, V, do1, lse1, delta1, S, heads, HW, HW, HW, dh, scale, out=dqkv[:, :C])
        ops.attn_bwd_dkv(Q, K, V, do1, lse1, delta1, S, heads, HW, HW, dh, scale, dK=dqkv[:, C:2 * C], dV=dqkv[:, 2 * C:])
        da1 = ops.gemm(dqkv, W[t + ".attn1.qkv:T"])
        dpin = ops.layernorm_bwd(cc("pin"), da1, W[t + ".norm1.weight"], cc("st1"), residual=dp1)
        dg = ops.gemm(dpin, W[p + ".proj_in.weight:T"])
        return ops.groupnorm_bwd(cc("x"), dg, S, HW, cfg.norm_groups, cs("gst"), W[p + ".norm.weight"],
                                 W[p + ".norm.bias"], False, residual=dout)

    # ------------------------------------------------------------------ backward
    def backward(self, stash: Stash, tap_grads: Sequence[Optional[torch.Tensor]]) -> torch.Tensor:
        """tap_grads: 9 fp16 tensors [S*s*s, C] (cond rows) in tap order.  Returns d loss / d x for the
        cond rows, fp16 [S*H*H, 8] (first 4 channels valid)."""
        assert self.need_backward and self.inject is None, "backward with injected attention is not supported"
        cfg, W = self.cfg, self.W
        rows, H = stash.misc["rows"], stash.misc["H"]
        S = rows // 2
        nb = len(cfg.block_out_channels)
        lpb = cfg.layers_per_block

        def add(a, b):
            if a is None:
                return b
            if b is None:
                return a
            return ops.axpby(a, b)

        # skip index bookkeeping: skips = [conv_in] + per down block (lpb resnet outs [+ downsample out])
        n_skips = 1 + sum(lpb + (1 if i < nb - 1 else 0) for i in range(nb))
        gskip: List[Optional[torch.Tensor]] = [None] * n_skips
        # ---- up path, reversed (blocks 2, 1, 0; block 3 does not feed any tap)
        cur = H                                  # spatial size of up block 2's output (after its upsampler)
        for _ in range(nb - 1 - 3):              # generic: taps end at up block 2
            pass
        # index of the skip consumed first by up block i: pops from the end
        pop_idx = n_skips
        consumed = {}
        for i in range(nb):
            for j in range(lpb + 1):
                pop_idx -= 1
                consumed[(i, j)] = pop_idx
        dh = tap_grads[8]
        cur = H
        plan = up_block_plan(cfg)
        for i in (2, 1, 0):
            if i < nb - 1:
                # upsampler backward: dgrad at the upsampled size, then 2x2 sum-pool
                p = f"up_blocks.{i}.upsamplers.0.conv"
                if UP2_DGRAD and (p + ".weight:ppT") in W:      # polyphase: one 4 x 4 stride-2 convolution, 16 instead of 36 tap-products
                    dh = ops.conv4x4s2(dh, W[p + ".weight:ppT"], S, cur, cur, W9T=W[p + ".weight:T"])
                    cur //= 2
                else:
                    du = ops.conv3x3(dh, W[p + ".weight:T"], S, cur, cur)
                    cur //= 2
                    dh = ops.sumpool2x2(du, S, cur, cur)
            for j in range(lpb, -1, -1):
                if i > 0:
                    dh = self._tr_bwd(f"up_blocks.{i}.attentions.{j}", dh, S, stash.tr[f"up_blocks.{i}.attentions.{j}"])
                dcat = self._res_bwd(f"up_blocks.{i}.resnets.{j}", dh, S, stash.res[f"up_blocks.{i}.resnets.{j}"])
                ch = plan[i][j][0]
                k = consumed[(i, j)]
                gskip[k] = add(gskip[k], dcat[:, ch:])
                dh = dcat[:, :ch]
            if i > 0:
                dh = add(dh, tap_grads[6 + i - 1])      # output of up block i-1 (after its upsampler)
        # ---- mid
        dh = add(dh, tap_grads[5])
        dh = self._res_bwd("mid_block.resnets.1", dh, S, stash.res["mid_block.resnets.1"])
        dh = add(dh, tap_grads[3])
        dh = self._tr_bwd("mid_block.attentions.0", dh, S, stash.tr["mid_block.attentions.0"])
        dh = add(dh, tap_grads[4])
        dh = self._res_bwd("mid_block.resnets.0", dh, S, stash.res["mid_block.resnets.0"])
        # ---- down path, reversed
        k = n_skips - 1
        for i in range(nb - 1, -1, -1):
            if i < nb - 1:
                dh = add(dh, gskip[k]); k -= 1
                if i < 3:
                    dh = add(dh, tap_grads[i])
                p = f"down_blocks.{i}.downsamplers.0.conv"
                dh = ops.conv3x3(dh, W[p + ".weight:T"], S, cur, cur, ops.CONV_S2T)
                cur *= 2
            for j in range(lpb - 1, -1, -1):
                dh = add(dh, gskip[k]); k -= 1
                if i < nb - 1:
                    dh = self._tr_bwd(f"down_blocks.{i}.attentions.{j}", dh, S,
                                      stash.tr[f"down_blocks.{i}.attentions.{j}"])
                dh = self._res_bwd(f"down_blocks.{i}.resnets.{j}", dh, S, stash.res[f"down_blocks.{i}.resnets.{j}"])
        dh = add(dh, gskip[0])
        return ops.conv3x3(dh, W["conv_in.weight:T"], S, H, H)

"""Injected attention (the reference's SatMixin / AttnModule step 1.5) on the libskg.so kernels.

  * CLIP-token variant      modules/clip_guided_attn.py:111-125
        s = sketch_proj(state); z = sketch_norm(cat([h, s], 1)); a = sketch_attn(z)[:, :N]; h += scale*conv1x1(a)
  * UNet-feature variant    modules/sketch_guided_attn.py:120-132
        z = sketch_norm(h); a = sketch_attn(z, encoder_hidden_states=res_sample); h += scale*conv1x1(a)

What does not depend on the latent is hoisted to ``set_state`` / ``set_res_samples`` (once per image): the
projected + normalised sketch tokens and, for the feature variant, the K / V projections of the residual
samples.  LayerNorm is per token, so LayerNorm(cat([h, s])) == cat(LayerNorm(h), LayerNorm(s)): the sketch
tokens are normalised once and copied into the tail of the per-step token buffer.

Buffers, CLIP variant, per block: K / V [rows][N + T_pad][2C] with T = 257 sketch tokens padded to a multiple of 8.
The sketch tokens' K / V rows do not depend on the latent and are written once per image; each step the image
tokens' K / V go straight into the first N slots of every batch row (one GEMM per row, no copy), the queries are
the N image tokens only (the reference slices the sketch-query outputs away, clip_guided_attn.py:119) and the
last padding keys are masked by Nkv < kv_stride.
"""
from __future__ import annotations

import os
from typing import Dict, List, Optional, Sequence

import torch

from . import ops
from .config import UNetConfig

SKETCH_TOKENS = 257
CLIP_DIM = 1024


def transformer_block_paths(cfg: UNetConfig) -> List[str]:
    """BasicTransformerBlock paths in unet.named_modules() order: down, up, mid (SatMixin.blocks order)."""
    nb = len(cfg.block_out_channels)
    out = []
    for i in range(nb - 1):
        out += [f"down_blocks.{i}.attentions.{j}.transformer_blocks.0" for j in range(cfg.layers_per_block)]
    for i in range(1, nb):
        out += [f"up_blocks.{i}.attentions.{j}.transformer_blocks.0" for j in range(cfg.layers_per_block + 1)]
    out.append("mid_block.attentions.0.transformer_blocks.0")
    return out


def block_dims(cfg: UNetConfig):
    boc = cfg.block_out_channels
    rev, rev_heads = tuple(reversed(boc)), tuple(reversed(cfg.num_heads))
    out = []
    for p in transformer_block_paths(cfg):
        parts = p.split(".")
        if parts[0] == "down_blocks":
            out.append((p, boc[int(parts[1])], cfg.num_heads[int(parts[1])]))
        elif parts[0] == "up_blocks":
            out.append((p, rev[int(parts[1])], rev_heads[int(parts[1])]))
        else:
            out.append((p, boc[-1], cfg.num_heads[-1]))
    return out


def module_name(block_path: str) -> str:
    """modules/clip_guided_attn.py:14-19: "sketch_attn." + path with '.' -> '_'."""
    return ("sketch_attn." + block_path).replace(".", "_")


def route_res_samples(res_samples: Sequence[Sequence[torch.Tensor]]) -> List[torch.Tensor]:
    """modules/sketch_guided_attn.py:29-40."""
    down, up = (), ()
    mid = (res_samples[-1][-1],)
    for layers in res_samples:
        if len(layers) == 3:
            down += (layers[0], layers[1])
            up += (layers[0], layers[1], layers[1])
    return list(down + up[::-1] + mid)


_ROWS_KV = os.environ.get("SKG_INJ_ROWS", "1") != "0"        # A/B switch: large maps, K / V of all batch rows in one launch
_BATCH_KV = os.environ.get("SKG_INJ_BATCH", "1") != "0"      # A/B switch (bench.py on one box)


class HipInjector:
    """Callable installed as ``HipUNet.inject``; ``variant`` is 'clip' or 'sketch'."""

    def __init__(self, cfg: UNetConfig, state_dict: Dict[str, torch.Tensor], variant: str, device):
        assert variant in ("clip", "sketch")
        self.cfg, self.variant, self.dev = cfg, variant, torch.device(device)
        self.scale = 1.0
        self.W: Dict[str, Dict[str, torch.Tensor]] = {}
        self.per_image: Dict[str, dict] = {}
        self.halves_equal = False      # set_res_samples: both CFG halves carry the same sketch features (checked, not assumed)
        h16 = lambda t: t.detach().to(self.dev, torch.float16).contiguous()
        for path, c, heads in block_dims(cfg):
            n = module_name(path)
            w = dict(C=c, heads=heads,
                     ng=h16(state_dict[f"{n}.sketch_norm.weight"]), nb=h16(state_dict[f"{n}.sketch_norm.bias"]),
                     wq=h16(state_dict[f"{n}.sketch_attn.to_q.weight"]),
                     wkv=h16(torch.cat([state_dict[f"{n}.sketch_attn.to_k.weight"],
                                        state_dict[f"{n}.sketch_attn.to_v.weight"]], 0)),
                     wo=h16(state_dict[f"{n}.sketch_attn.to_out.0.weight"]),
                     bo=h16(state_dict[f"{n}.sketch_attn.to_out.0.bias"]),
                     wc=h16(state_dict[f"{n}.sketch_conv.weight"].reshape(c, c)),
                     bc=h16(state_dict[f"{n}.sketch_conv.bias"]))
            if variant == "clip":
                w["wqkv"] = torch.cat([w["wq"], w["wkv"]], 0).contiguous()
                w["wp"] = h16(state_dict[f"{n}.sketch_proj.weight"])
                w["bp"] = h16(state_dict[f"{n}.sketch_proj.bias"])
            self.W[path] = w

    def set_scale(self, scale: float):
        self.scale = float(scale)

    # ---- once per image ---------------------------------------------------------------------------------
    def set_state(self, sketch_state: Optional[torch.Tensor]):
        """CLIP variant.  sketch_state [rows, 257, 1024] (uncond rows zero, modules/clip_guided_inf.py:107)."""
        self.per_image = {}
        if sketch_state is None:
            return
        rows, T, D = sketch_state.shape
        st = sketch_state.to(self.dev, torch.float16).reshape(rows * T, D).contiguous()
        for path, w in self.W.items():
            s = ops.gemm(st, w["wp"], bias=w["bp"])
            zs = ops.layernorm(s, w["ng"], w["nb"])
            self.per_image[path] = dict(kvs=ops.gemm(zs, w["wkv"]), rows=rows, T=T, bufs={})

    def set_res_samples(self, res_samples: Optional[Sequence[Sequence[torch.Tensor]]]):
        """Feature variant.  res_samples: per down block a tuple of NCHW tensors [rows, C, h, w]."""
        self.per_image = {}
        self.halves_equal = False
        if res_samples is None:
            return
        routed = route_res_samples(res_samples)
        # the SketchEncoder usually sees the same sketch for the uncond and the cond row of a sample: when the two halves of
        # EVERY routed tensor are equal (checked here, once per image) the injected attention is text-independent too and
        # HipUNet's shared CFG front extends through it
        self.halves_equal = all(r.shape[0] % 2 == 0 and torch.equal(r[: r.shape[0] // 2], r[r.shape[0] // 2:]) for r in routed)
        for (path, w), r in zip(self.W.items(), routed):
            rows, C, hh, ww = r.shape
            assert C == w["C"]
            tok = r.to(self.dev, torch.float16).permute(0, 2, 3, 1).reshape(rows * hh * ww, C).contiguous()
            kv = ops.gemm(tok, w["wkv"])                       # res_sample is used un-normalised (:127)
            self.per_image[path] = dict(K=kv[:, :C], V=kv[:, C:], rows=rows, N=hh * ww)

    # ---- per UNet evaluation ------------------------------------------------------------------------------
    def __call__(self, path: str, h, rows: int, N: int, heads: int, cond_only: bool = False, out=None):
        """cond_only (sketch variant, halves_equal): h holds the cond rows only (rows // 2 of them); K / V of the cond half.
        h (and out) may be ops.Pair objects - HipUNet's accuracy mode: the stream enters sketch_norm as hi + lo and the scaled
        injection is added to the pair in fp32 (the result is a pair again)."""
        pair = isinstance(h, ops.Pair)
        pi = self.per_image.get(path)
        if pi is None:
            if out is not None:
                for src, dst in ((h.hi, out.hi), (h.lo, out.lo)) if pair else ((h, out),):
                    ops.batch_copy(src, src.shape[0], dst, src.shape[0], 1, src.shape[0])
                return out
            return h
        if pair:
            hp, h = h, h.hi
            res = out or ops.Pair.empty(h.shape[0], h.shape[1], self.dev)
            norm = lambda g, b: ops.layernorm_hilo(hp.hi, hp.lo, g, b)
            # h + scale * conv1x1(o) in fp32 on the pair
            last = lambda o, wc, bc: (ops.gemm(o, wc, out=res.hi, out_lo=res.lo, bias=bc, residual=hp.hi, residual_lo=hp.lo,
                                               alpha=self.scale), res)[1]
        else:
            norm = lambda g, b: ops.layernorm(h, g, b)
            last = lambda o, wc, bc: ops.gemm(o, wc, out, bias=bc, residual=h, alpha=self.scale)
        w = self.W[path]
        C = w["C"]
        dh = C // heads
        scale = dh ** -0.5
        if self.variant == "sketch":
            assert pi["rows"] == rows and pi["N"] == N
            K, V, r = pi["K"], pi["V"], rows
            if cond_only:
                assert self.halves_equal
                r = rows // 2
                K, V = K[r * N:], V[r * N:]
            z = norm(w["ng"], w["nb"])
            q = ops.gemm(z, w["wq"])
            a = ops.attn_fwd(q, K, V, r, heads, N, N, N, dh, scale, v_rows=True)
            o = ops.gemm(a, w["wo"], bias=w["bo"])
            return last(o, w["wc"], w["bc"])
        # CLIP variant: self-attention of the N image-token queries over [N image tokens ; T sketch tokens]
        assert pi["rows"] == rows
        T = pi["T"]
        L = (N + T + 7) // 8 * 8
        kvbuf = pi["bufs"].get(N)
        if kvbuf is None:
            kvbuf = torch.zeros(rows * L, 2 * C, device=self.dev, dtype=torch.float16)
            ops.batch_copy(pi["kvs"], T, kvbuf[N:], L, rows, T)     # K / V of the normalised sketch tokens, once per image
            pi["bufs"][N] = kvbuf
        zh = norm(w["ng"], w["nb"])
        # K / V of the image tokens go to rows [b L, b L + N) of the [rows * L, 2C] buffer (the sketch tokens sit behind them).
        # Large maps: one GEMM per batch row straight into place (each launch fills the chip).  Small maps - a batch row is
        # under 256 tiles of 128 x 160 - : ONE GEMM over all rows for q, k and v into a dense buffer + one strided copy of the
        # k | v columns, 2 launches instead of 1 + `rows` launches of ~10-25 us each (round 3: ~12 800 of config 5's 16 000
        # GEMM launches per batch were these; same-box A/B 2.218 -> 2.266 images/s)
        if _BATCH_KV and rows > 2 and ((N + 127) // 128) * ((2 * C + 159) // 160) < 256:
            dense = ops.gemm(zh, w["wqkv"])
            q = dense[:, :C]
            ops.batch_copy(dense[:, C:], N, kvbuf, L, rows, N)
        else:
            q = ops.gemm(zh, w["wq"])
            if _ROWS_KV and C % 64 == 0:      # one launch: row block b of the product -> rows [b L, b L + N) of the K / V buffer
                ops.gemm_rows(zh, w["wkv"], kvbuf, N, L)
            else:
                for b in range(rows):
                    ops.gemm(zh[b * N:(b + 1) * N], w["wkv"], out=kvbuf[b * L:b * L + N])
        a = ops.attn_fwd(q, kvbuf[:, :C], kvbuf[:, C:], rows, heads, N, N + T, L, dh, scale, v_rows=True)
        o = ops.gemm(a, w["wo"], bias=w["bo"])
        return last(o, w["wc"], w["bc"])

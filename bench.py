#!/usr/bin/env python3
"""Benchmark of the sketch-guided sampler hot path (BASELINE.json metric).

One "step" = one complete pass of the hot path over one batch of independent samples on every GPU:
50 DDIM steps (CFG 7.5) from resident latents to decoded uint8 images (VAE decode on the rank that sampled,
then - N > 1 - the gather of the decoded images on rank 0).  ``--config`` picks the BASELINE.json workload:

    2 (default)  configs[1]: SD1.5, 8 samples per GPU, 512x512, LGP sketch guidance on steps 0..25
                 (configs[2] is the same workload at --gpus 8)
    4            configs[3]: SD1.5, 8 samples per GPU, 512x512, sketch_guided_attn injection, no LGP gradient
    5            configs[4]: SD2.1 architecture, 4 samples per GPU, 768x768, clip_guided_attn injection

``value`` is timed in the ACCURACY mode of the UNet (``residual_fp32``: the mode that meets north_star's
"<= 1e-3 max latent-eps deviation vs reference"; round 6 - VERDICT r5 next #1); the all-fp16 mode (the reference's own
GPU configuration, app.py:34) is timed beside it on three batches and reported as ``config.fast_fp16_value``
(``--fast-fp16`` swaps the two).  Everything a reader needs to judge the line is a FLAT scalar: ``config.mode``,
``config.eps_max`` / ``eps_bound`` / ``eps_rel`` / ``eps_max_unit_var``, ``config.fast_fp16_value``, the box calibration
``config.box_mfma_tflops`` / ``box_sclk_mhz`` / ``box_power_w`` / ``box_power_cap_w`` and ``roofline.*_frac``.

Weights, text embeddings, sketch inputs and initial latents are synthetic (seeded) and resident in HBM before
the timed region.  N > 1: one process per GPU (torch.distributed / RCCL), rank 0's weights are broadcast
once, every rank samples its own images (weak scaling, no per-step collective).

    python bench.py --gpus 1 --steps 2 --warmup 1
    python bench.py --gpus 8                       (launches itself under torch.distributed.run, one rank per GPU)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 ... bench.py --gpus 8

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` (dominant contraction kernel,
HIP-event timed in an extra instrumented pass; `rocprof` = the committed rocprofv3 average of the same
kernel on the same command, when profiles/ holds one) and `cpu_baseline` (the CPU oracle timed on this box's
host cores on a bounded sample; baseline only).
"""
import argparse
import csv
import glob
import json
import os
import re
import socket
import subprocess
import sys
import threading
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

PEAK_FP16_TFLOPS = 2500.0     # dense fp16 MFMA peak, MI355X_MICROARCH.md
PEAK_HBM_TBS = 8.0            # HBM3E peak (spec), MI355X_MICROARCH.md

# algorithmic GFLOP per UNet row evaluation
GF_SD15_FWD, GF_SD15_BWD, GF_LGP = 803.27, 929.33, 40.50
GF_C4_ROW = 803.27 + 186.46       # + sketch_guided_attn injection
GF_C5_ROW = 2149.1 + 852.8        # SD2.1 @ 96x96 + clip_guided_attn injection


def f_img_tflop(config: int, T: int) -> float:
    """Algorithmic TFLOP per image.  Config 2, T = 50 -> 107.65; config 4 -> 98.97; config 5 -> 300.19."""
    if config == 2:
        guided = sum(1 for i in range(T) if not (i > 0.5 * T))
        return (T * 2 * GF_SD15_FWD + guided * (GF_SD15_BWD + 3 * GF_LGP)) / 1e3
    return T * 2 * (GF_C4_ROW if config == 4 else GF_C5_ROW) / 1e3


WORKLOADS = {      # (<= 120 characters: parsers of the contract line cut longer strings; the long form is `workload_detail`)
    2: "BASELINE configs[1]: SD1.5 fp16, {S} samples/GPU, 512x512, {T} {sched} steps, CFG 7.5, LGP guidance steps 0..{G}",
    4: "BASELINE configs[3]: SD1.5 fp16, {S} samples/GPU, 512x512, {T} {sched} steps, CFG 7.5, sketch_guided_attn",
    5: "BASELINE configs[4]: SD2.1 fp16, {S} samples/GPU, 768x768, {T} {sched} steps, CFG 7.5, clip_guided_attn",
}
WORKLOAD_DETAILS = {
    2: "BASELINE.json configs[1]: SD1.5 architecture (synthetic seeded weights), {S} independent samples per GPU, "
       "512x512 (64x64 latents), {T} {sched} steps, CFG 7.5, LGP sketch guidance on steps 0..{G} (beta 1.6)",
    4: "BASELINE.json configs[3]: SD1.5 architecture (synthetic seeded weights), {S} independent samples per GPU, "
       "512x512 (64x64 latents), {T} {sched} steps, CFG 7.5, sketch_guided_attn injection (scale 1.0, synthetic "
       "res_samples), no LGP gradient",
    5: "BASELINE.json configs[4]: SD2.1 architecture (head_dim 64, 1024-wide context, synthetic seeded weights), {S} "
       "independent samples per GPU, 768x768 (96x96 latents), {T} {sched} steps, CFG 7.5, clip_guided_attn injection on "
       "[zeros; 257 CLIP tokens] (scale 1.0)",
}
EPS_BOUND = 1e-3      # north_star: "<= 1e-3 max latent-eps deviation vs reference"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=2, choices=(2, 4, 5))
    ap.add_argument("--samples-per-gpu", type=int, default=None, help="default 8 (configs 2, 4) / 4 (config 5)")
    ap.add_argument("--ddim-steps", type=int, default=50)
    ap.add_argument("--scheduler", default="ddim", choices=("ddim", "dpm"),
                    help="dpm = DPM-Solver++ 2M as app.py:13-25 ships (the BASELINE metric is DDIM)")
    ap.add_argument("--gather", default="images", choices=("images", "latents"),
                    help="images (default): VAE decode on-rank + gather of uint8 images inside the timed region, as "
                         "north_star states; latents: gather the fp32 latents, no decode (round-1 behaviour)")
    ap.add_argument("--no-guidance", action="store_true", help="informational: config 2 without the LGP guidance (no backward)")
    ap.add_argument("--mode", default="tolerance", choices=("tolerance", "fast"),
                    help="what `value` is timed in: tolerance (default) = HipUNet(residual_fp32=True), the mode that meets north_star's "
                         "1e-3 eps bound; fast = every stored tensor fp16 (the reference's own GPU configuration)")
    ap.add_argument("--residual-fp32", dest="mode", action="store_const", const="tolerance", help="= --mode tolerance")
    ap.add_argument("--fast-fp16", dest="mode", action="store_const", const="fast", help="= --mode fast")
    ap.add_argument("--no-second-mode", "--no-at-tolerance", dest="no_second_mode", action="store_true",
                    help="skip the second timed region (the same workload in the OTHER mode: `config.fast_fp16_value`)")
    ap.add_argument("--second-mode-steps", "--at-tolerance-steps", dest="second_mode_steps", type=int, default=3,
                    help="timed batches of the second region (after 1 warm-up)")
    ap.add_argument("--second-mode-multi", "--at-tolerance-multi", dest="second_mode_multi", action="store_true",
                    help="run the second region at --gpus > 1 too (default: N = 1 only - it builds a second full workload on every rank)")
    ap.add_argument("--no-box-probe", action="store_true", help="skip the box calibration (MFMA probe + rocm-smi sample)")
    ap.add_argument("--plumbing-check", action="store_true",
                    help="tests only: launcher / rendezvous / barrier / JSON relay without the hot path (runs without a GPU)")
    ap.add_argument("--graph", action="store_true", help="replay the two step variants from captured hipGraphs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--shape-report", default=None, help="write a per-shape table of the GEMM / conv / attention launches of one batch")
    ap.add_argument("--first-sample", type=int, default=0, help="global index of rank 0's first sample (seeds depend on the global index only)")
    ap.add_argument("--dump-images", default=None, help="rank 0 saves the gathered result of the LAST timed batch (torch.save)")
    return ap.parse_args()


ATTN_NAMES = {16: "1, 1, 2", 32: "1, 2, 2", 40: "2, 3, 2", 64: "2, 4, 2", 80: "3, 5, 2", 160: "5, 10, 1"}
ATTN_SHORT_NAMES = {40: "2, 3", 64: "2, 4", 80: "3, 5", 160: "5, 10"}      # attn_fwd_short_kernel<KS, ND>: kv_stride <= 80


class LaunchTimer:
    """Times every skg_gemm_f16 / skg_conv3x3_f16 / skg_attn_fwd launch with HIP events on the launch stream."""

    def __init__(self, ops):
        self.ops, self.rec = ops, []
        self._gemm, self._conv, self._attn, self._keep = ops.gemm, ops.conv3x3, ops.attn_fwd, ops.gemm_geglu_keep
        self._up2, self._c4, self._ffb, self._xab = ops.conv_up2, ops.conv4x4s2, ops.ff_block, ops.xattn_block
        self._ffp = ops.ff_block_proj
        self._csc = ops.conv3x3_sc
        self._up2p = ops.conv_up2_pairout
        self._wino = ops.conv3x3_wino

    def __enter__(self):
        from sketch2img_amd._lib import lib
        ops = self.ops

        MODES = {"DIRECT": 0, "S1": 1, "S2": 2, "UP2": 3, "S2T": 4}

        def kname(variant, mode, gn=False, hilo=False):
            """The name rocprofv3 prints for the instantiation that ran (template arguments spelled out); gn: the one
            whose epilogue also writes the GroupNorm partial sums of the output; hilo: the accuracy mode's pair epilogue."""
            bn = variant % 1000
            g, h = "true" if gn else "false", "true" if hilo else "false"
            if variant in (8160, 8320):
                return f"gemm8_kernel<{variant - 8000}, {MODES[mode]}, 0, {g}, {h}>"
            if variant >= 2000:
                stages = 3 if variant >= 10000 else 2
                return f"gemm2_kernel<{256 if bn == 320 else 128}, {bn}, 2, {4 if bn == 320 else 2}, {MODES[mode]}, {stages}, {g}, {h}>"
            return f"gemm_kernel<{bn}, {MODES[mode]}>"

        def is_pair(k):
            return k.get("out_lo") is not None

        def ev():
            return torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)

        def gemm(A, B, *a, **k):
            M, K = A.shape
            N = B.shape[0]
            e0, e1 = ev()
            e0.record()
            out = self._gemm(A, B, *a, **k)
            e1.record()
            nbytes = 2.0 * (M * K + N * K + M * N * (2 if k.get("out_f32") else 1) + (M * N if k.get("residual") is not None else 0))
            tag = "+res" * (k.get("residual") is not None) + "+geglu" * bool(k.get("geglu")) + "+f32" * bool(k.get("out_f32"))
            gs = k.get("gn_stats")
            gn = gs is not None and bool(lib.skg_gemm_gn_fused(M, N, K, 0, 0, gs[0], gs[1]))
            nbytes += 2.0 * M * N * (is_pair(k) + (k.get("residual_lo") is not None))
            tag += "+pair" * is_pair(k)
            self.rec.append((kname(lib.skg_gemm_variant(M, N, K, 0, 0), "DIRECT", gn, is_pair(k)), 2.0 * M * N * K, e0, e1, nbytes,
                             f"gemm M{M} N{N} K{K}{tag}"))
            return out

        def conv(X, Wp, rows, IH, IW, mode=0, *a, **k):
            Cin, Cout = X.shape[1], Wp.shape[0]
            OH = IH if mode == 0 else (IH // 2 if mode == 1 else IH * 2)
            M = rows * OH * OH
            e0, e1 = ev()
            e0.record()
            out = self._conv(X, Wp, rows, IH, IW, mode, *a, **k)
            e1.record()
            name = ("S1", "S2", "UP2", "S2T")[mode]
            nbytes = 2.0 * (X.shape[0] * Cin + Cout * 9 * Cin + M * Cout + (M * Cout if k.get("residual") is not None else 0))
            gg = k.get("gn_groups")
            gn = gg is not None and bool(lib.skg_gemm_gn_fused(M, Cout, 9 * Cin, Cin, 1 + mode, OH * OH, gg))
            nbytes += 2.0 * M * Cout * (is_pair(k) + (k.get("residual_lo") is not None))
            self.rec.append((kname(lib.skg_gemm_variant(M, Cout, 9 * Cin, Cin, 1 + mode), name, gn, is_pair(k)), 2.0 * M * Cout * 9 * Cin, e0, e1, nbytes,
                             f"conv {name} M{M} Cin{Cin} Cout{Cout}" + "+res" * (k.get("residual") is not None) + "+pair" * is_pair(k)))
            return out

        def conv_sc(X, X2, Wcat, rows, IH, IW, *a, **k):     # conv2 + the 1x1 shortcut of a ResnetBlock as one implicit GEMM
            Cin, K2, Cout = X.shape[1], X2.shape[1], Wcat.shape[0]
            M, K = rows * IH * IW, 9 * Cin + K2
            e0, e1 = ev()
            e0.record()
            out = self._csc(X, X2, Wcat, rows, IH, IW, *a, **k)
            e1.record()
            gg = k.get("gn_groups")
            gn = gg is not None and bool(lib.skg_gemm_gn_fused(M, Cout, K, Cin, 1, IH * IW, gg))
            self.rec.append((kname(lib.skg_gemm_variant(M, Cout, K, Cin, 1), "S1", gn, is_pair(k)), 2.0 * M * Cout * K, e0, e1,
                             2.0 * (M * Cin + M * K2 + Cout * K + M * Cout * (1 + is_pair(k))),
                             f"conv S1 + shortcut M{M} Cin{Cin} K2 {K2} Cout{Cout}" + "+pair" * is_pair(k)))
            return out

        def conv_wino(X, U, rows, IH, IW, *a, **k):      # Winograd F(2x2, 3x3): input transform + split GEMM + output transform (three launches)
            Cin, Cout = U.shape[1] // 16, U.shape[0]      # (X is None when the GroupNorm in front wrote the input transform: V=)
            M = rows * IH * IW
            e0, e1 = ev()
            e0.record()
            out = self._wino(X, U, rows, IH, IW, *a, **k)
            e1.record()
            # ALGORITHMIC flops of the convolution (the path executes 16 / 36 of them) and bytes (X, U, Y; V and the fp32 slabs are the path's own traffic)
            self.rec.append(("wino_conv3x3 (wino_in | gn_small_wino + gemm2 split x16 + wino_out)", 2.0 * M * Cout * 9 * Cin, e0, e1,
                             2.0 * (M * Cin + Cout * 16 * Cin + M * Cout + (M * Cout if k.get("residual") is not None else 0)),
                             f"conv S1 Winograd M{M} Cin{Cin} Cout{Cout}" + "+res" * (k.get("residual") is not None), 3))
            return out

        def v2name(M, N, K, Cin, mode, label, phases=1):
            """gemm2.hip instantiation of a polyphase launch (gemm8.hip declines them; `phases` grids in one launch)."""
            var = lib.skg_gemm_variant(M, N, K, Cin, 1 + mode)
            bn = var % 1000 if 2000 <= var < 8000 or var >= 10000 else 160
            three = var >= 10000 and phases == 1
            return f"gemm2_kernel<128, {bn}, 2, 2, {MODES[label]}, {3 if three else 2}, false, false>"

        def conv_up2(X, Wpp, rows, IH, IW, *a, **k):
            # EXECUTED flops: four 4-tap convolutions over the low-res map (the 9-tap form of the same layer: 36 tap-products)
            Cin, Cout = X.shape[1], Wpp.shape[1]
            M = rows * IH * IW
            e0, e1 = ev()
            e0.record()
            out = self._up2(X, Wpp, rows, IH, IW, *a, **k)
            e1.record()
            one = ((M + 127) // 128) * ((Cout + 159) // 160) < 200            # the four phases as one grid (gemm.hip)
            self.rec.append((v2name(M, Cout, 4 * Cin, Cin, 0, "S1", 4 if one else 1), 2.0 * M * Cout * 16 * Cin, e0, e1,
                             2.0 * (M * Cin + 16 * Cout * Cin + 4 * M * Cout), f"conv UP2 polyphase M{4 * M} Cin{Cin} Cout{Cout}", 1 if one else 4))
            return out

        def conv_up2_pairout(X, Wpp, rows, IH, IW, *a, **k):      # accuracy mode: the same polyphase launches with a pair output
            Cin, Cout = X.shape[1], Wpp.shape[1]
            M = rows * IH * IW
            e0, e1 = ev()
            e0.record()
            out = self._up2p(X, Wpp, rows, IH, IW, *a, **k)
            e1.record()
            one = ((M + 127) // 128) * ((Cout + 159) // 160) < 200
            self.rec.append((v2name(M, Cout, 4 * Cin, Cin, 0, "S1", 4 if one else 1).replace("false>", "true>"), 2.0 * M * Cout * 16 * Cin, e0, e1,
                             2.0 * (M * Cin + 16 * Cout * Cin + 8 * M * Cout), f"conv UP2 polyphase M{4 * M} Cin{Cin} Cout{Cout}+pair", 1 if one else 4))
            return out

        def conv4x4s2(X, W16, rows, IH, IW, *a, **k):
            Cin, Cout = X.shape[1], W16.shape[0]
            M = rows * (IH // 2) * (IW // 2)
            e0, e1 = ev()
            e0.record()
            out = self._c4(X, W16, rows, IH, IW, *a, **k)
            e1.record()
            self.rec.append((v2name(M, Cout, 16 * Cin, Cin, 1, "S2"), 2.0 * M * Cout * 16 * Cin, e0, e1,
                             2.0 * (4 * M * Cin + 16 * Cout * Cin + M * Cout), f"conv 4x4 S2 (UP2 dgrad) M{M} Cin{Cin} Cout{Cout}"))
            return out

        def attn(Q, K, Vt, batch, heads, Nq, Nkv, kv_stride, dh, scale, *a, **k):
            e0, e1 = ev()
            e0.record()
            out = self._attn(Q, K, Vt, batch, heads, Nq, Nkv, kv_stride, dh, scale, *a, **k)
            e1.record()
            fl = 4.0 * batch * heads * Nq * Nkv * dh                      # QK^T + PV
            nbytes = 2.0 * batch * heads * dh * (2 * Nq + 2 * Nkv)
            vrow = "true" if k.get("v_rows") else "false"         # row-major V through the LDS transpose read
            if k.get("v_rows") and kv_stride <= 80 and dh in ATTN_SHORT_NAMES and not os.environ.get("SKG_NO_ATTN_SHORT"):
                name = f"attn_fwd_short_kernel<{ATTN_SHORT_NAMES[dh]}>"      # the 77 text tokens: the LDS-resident kernel
            else:
                name = f"attn_fwd_kernel<{ATTN_NAMES.get(dh, '?')}, false, {14 if dh == 64 else 0}, {vrow}>"
                if dh == 40 and -(-Nq // 192) * heads * batch >= 1024 and not os.environ.get("SKG_ATTN_VAR"):
                    name = f"attn_fwd_kernel<2, 3, 3, false, 14, {vrow}>"      # three query tiles per wave (attention.hip attn_fwd_launch_qt)
            self.rec.append((name, fl, e0, e1, nbytes, f"attn B{batch} H{heads} Nq{Nq} Nkv{Nkv} d{dh}"))
            return out

        def gemm_keep(A, B, *a, **k):        # FF1 with the fused gate that also stores the pre-activation
            M, K = A.shape
            N = B.shape[0]
            e0, e1 = ev()
            e0.record()
            out = self._keep(A, B, *a, **k)
            e1.record()
            self.rec.append((kname(lib.skg_gemm_variant(M, N, K, 0, 0), "DIRECT"), 2.0 * M * N * K, e0, e1,
                             2.0 * (M * K + N * K + M * N + M * N // 2), f"gemm M{M} N{N} K{K}+geglu+keep"))
            return out

        def hi(X):
            return X.hi if isinstance(X, ops.Pair) else X

        def ff_block(X, gamma, beta, eps, pack, *a, **k):       # norm3 + FF1 + gate + FF2 + residual in one launch
            M, C = hi(X).shape
            Fh = pack.shape[0] * 32
            e0, e1 = ev()
            e0.record()
            out = self._ffb(X, gamma, beta, eps, pack, *a, **k)
            e1.record()
            self.rec.append(("ff_block_kernel<10>", 2.0 * M * C * 3 * Fh, e0, e1, 2.0 * (2 * M * C + 3 * C * Fh),
                             f"ff_block M{M} C{C} F{Fh} (LN + FF1 + gate + FF2 + res)" + "+keep" * (k.get("keep_from") is not None)))
            return out

        def ff_block_proj(X, gamma, beta, eps, pack, *a, **k):  # ... + proj_out + outer residual (five chunks more in the pack)
            M, C = hi(X).shape
            Fh = (pack.shape[0] - 5) * 32
            e0, e1 = ev()
            e0.record()
            out = self._ffp(X, gamma, beta, eps, pack, *a, **k)
            e1.record()
            self.rec.append(("ff_block_kernel<10, proj>", 2.0 * M * C * (3 * Fh + C), e0, e1, 2.0 * (3 * M * C + 3 * C * Fh + C * C),
                             f"ff_block M{M} C{C} F{Fh} (LN + FF1 + gate + FF2 + res + proj_out + res)" + "+keep" * (k.get("keep_from") is not None)))
            return out

        def xattn_block(X, HW, heads, Nkv, *a, **k):            # norm2 + to_q + text attention + to_out + residual in one launch
            M, C = hi(X).shape
            e0, e1 = ev()
            e0.record()
            out = self._xab(X, HW, heads, Nkv, *a, **k)
            e1.record()
            # algorithmic flops (unpadded head width): to_q + to_out + QK^T + PV
            self.rec.append(("xattn_block_kernel", 4.0 * M * C * C + 4.0 * M * Nkv * C, e0, e1,
                             2.0 * (2 * M * C + 2 * C * C + 2 * (M // HW) * Nkv * C),
                             f"xattn_block M{M} C{C} H{heads} Nkv{Nkv} (LN + to_q + attention + to_out + res)"))
            return out

        ops.gemm, ops.conv3x3, ops.attn_fwd, ops.gemm_geglu_keep = gemm, conv, attn, gemm_keep
        ops.conv_up2, ops.conv4x4s2, ops.ff_block, ops.xattn_block = conv_up2, conv4x4s2, ff_block, xattn_block
        ops.ff_block_proj = ff_block_proj
        ops.conv3x3_sc = conv_sc
        ops.conv_up2_pairout = conv_up2_pairout
        ops.conv3x3_wino = conv_wino
        return self

    def __exit__(self, *exc):
        self.ops.gemm, self.ops.conv3x3, self.ops.attn_fwd, self.ops.gemm_geglu_keep = self._gemm, self._conv, self._attn, self._keep
        self.ops.conv_up2, self.ops.conv4x4s2, self.ops.ff_block, self.ops.xattn_block = self._up2, self._c4, self._ffb, self._xab
        self.ops.ff_block_proj = self._ffp
        self.ops.conv3x3_sc = self._csc
        self.ops.conv_up2_pairout = self._up2p
        self.ops.conv3x3_wino = self._wino

    def summary(self):
        """per kernel: [launches, flops, seconds, algorithmic bytes, roofline seconds, seconds of HBM-bound launches];
        a launch's roofline time = max(flops / MFMA peak, algorithmic bytes / HBM peak)."""
        torch.cuda.synchronize()
        agg = {}
        for name, fl, e0, e1, nb, _, *nl in self.rec:      # (nl: kernel launches behind one record - the four polyphase grids)
            a = agg.setdefault(name, [0, 0.0, 0.0, 0.0, 0.0, 0.0])
            t = e0.elapsed_time(e1) * 1e-3
            t_m, t_h = fl / (PEAK_FP16_TFLOPS * 1e12), nb / (PEAK_HBM_TBS * 1e12)
            a[0] += nl[0] if nl else 1; a[1] += fl; a[2] += t; a[3] += nb; a[4] += max(t_m, t_h); a[5] += t if t_h > t_m else 0.0
        return agg

    def shape_report(self, path):
        """Per-shape table (launches, total ms, avg us, TFLOP/s, algorithmic TB/s), sorted by total time."""
        torch.cuda.synchronize()
        agg = {}
        for name, fl, e0, e1, nb, shape, *_ in self.rec:
            a = agg.setdefault((shape, name), [0, 0.0, 0.0, 0.0])
            a[0] += 1; a[1] += fl; a[2] += e0.elapsed_time(e1) * 1e-3; a[3] += nb
        rows = sorted(agg.items(), key=lambda kv: -kv[1][2])
        tot = sum(v[2] for _, v in rows)
        with open(path, "w") as f:
            f.write(f"GEMM / conv / attention launches of one batch, HIP events around each launch: {tot * 1e3:.1f} ms in {len(self.rec)} launches\n")
            f.write(f"{'shape':48s} {'kernel':36s} {'n':>6s} {'ms':>8s} {'avg us':>8s} {'TF/s':>7s} {'TB/s':>6s} {'%':>5s}\n")
            for (shape, name), (n, fl, sec, nb) in rows:
                f.write(f"{shape:48s} {name:36s} {n:6d} {sec * 1e3:8.2f} {sec / n * 1e6:8.1f} {fl / sec / 1e12:7.0f} "
                        f"{nb / sec / 1e12:6.2f} {100 * sec / tot:5.1f}\n")


def by_operator(agg):
    """{operator: launches, seconds, TFLOP/s, fraction of the MFMA peak} from the per-instantiation table."""
    def op(name):
        if name.startswith("attn"):
            return "attention_fwd"
        if name.startswith("ff_block"):
            return "ff_block"
        if name.startswith("xattn_block"):
            return "xattn_block"
        if name.startswith("wino"):
            return "conv3x3"
        args = name[name.index("<") + 1:-1].split(", ")
        mode = int(args[4]) if name.startswith("gemm2") else int(args[1])
        return {0: "gemm", 1: "conv3x3"}.get(mode, "conv3x3_resample")
    out = {}
    for k, v in agg.items():
        o = out.setdefault(op(k), [0, 0.0, 0.0, []])
        o[0] += v[0]; o[1] += v[1]; o[2] += v[2]; o[3].append(k)
    return {k: dict(launches=v[0], seconds=v[2], tflops=v[1] / v[2] / 1e12, frac=v[1] / v[2] / 1e12 / PEAK_FP16_TFLOPS,
                    kernels=sorted(v[3])) for k, v in sorted(out.items(), key=lambda kv: -kv[1][2])}


# ------------------------------------------------------------------------------- committed rocprofv3 evidence
_PROFILE_TAG = ""        # "fast" when `value` is timed in the all-fp16 mode (profiles/r<NN>fast_cfg<C>_*: tools/collect_profiles.sh)


def _latest_profile(config: int, suffix: str):
    """profiles/r<NN>[fast]_cfg<config>_<suffix> of the latest round that has one (round-1 files carry no cfg tag).  Rounds 1-5 profiled the
    all-fp16 mode under the plain tag; from round 6 on the plain tag is the accuracy mode (the headline) and `fast` the all-fp16 one."""
    cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r[0-9][0-9]{_PROFILE_TAG}_cfg{config}_{suffix}")))
    if _PROFILE_TAG == "fast":      # (rounds <= 5: the plain tag WAS the all-fp16 mode)
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", f"r0[1-5]_cfg{config}_{suffix}"))) + cands
    else:
        cands = [c for c in cands if os.path.basename(c) >= "r06"] or sorted(glob.glob(os.path.join(ROOT, "profiles", f"r0[3-5]acc_cfg{config}_{suffix}")))
    if not cands and config == 2:
        legacy = {"hbm_counters.json": "r01_hbm_counters.json", "kernel_stats.csv": "r01_kernel_stats_final.csv"}
        p = os.path.join(ROOT, "profiles", legacy.get(suffix, "-"))
        cands = [p] if os.path.exists(p) else []
    return cands[-1] if cands else None


def _norm_kernel(name: str) -> str:
    name = name.replace("(anonymous namespace)::", "")
    m = re.match(r"^(void )?([\w:]+(<[^(]*>)?)", name)
    return m.group(2) if m else name


def pmc_traffic(config: int, kernel_name: str):
    """HBM bytes per launch of `kernel_name` from the COMMITTED rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE
    collected in separate passes over the same workload, tools/collect_profiles.sh).  Units are KiB; on gfx950
    FETCH_SIZE counts half the bytes of wide coalesced reads (MI355X_MICROARCH.md section HBM; re-checked here on
    GEGLU / GroupNorm-apply whose read:write byte ratio is known) -> doubled.  A constant of the committed
    profile, not a measurement of this run."""
    path = _latest_profile(config, "hbm_counters.json")
    if path is None:
        return None, None
    for k, v in json.load(open(path)).items():
        if kernel_name in k.replace("void ", "") and "FETCH_SIZE" in v and "WRITE_SIZE" in v:
            f, w = v["FETCH_SIZE"], v["WRITE_SIZE"]
            return (2.0 * f["sum"] / f["launches"] + w["sum"] / w["launches"]) * 1024.0, os.path.relpath(path, ROOT)
    return None, os.path.relpath(path, ROOT)


def rocprof_duration(config: int, kernel_name: str):
    """(average ns, launches, file) of `kernel_name` in the committed rocprofv3 --kernel-trace --stats summary."""
    path = _latest_profile(config, "kernel_stats.csv")
    if path is None:
        return None
    for r in csv.DictReader(open(path)):
        if _norm_kernel(r["Name"]) == kernel_name:
            return float(r["AverageNs"]), int(r["Calls"]), os.path.relpath(path, ROOT)
    return None


# ---------------------------------------------------------------------------------------------- CPU baseline
def cpu_baseline(extrapolate_c2: bool = True):
    """The CPU oracle (a port of the reference's formulation: eager fp32, autograd through BOTH CFG rows,
    materialised 9320-channel tensor, modules/pipeline.py:83-161) on the host cores, as SURVEY 8(d) defines it:
    BASELINE config[0] - one sketch, 256x256 (32x32 latents), 10 DDIM steps, guidance on steps 0..5 - 1 warm-up run +
    3 timed runs.  Second, labelled figure: one guided + one unguided step of ONE config-2 sample (512x512) after a
    warm-up evaluation, extrapolated to 26 + 24 steps."""
    from oracle import ddim as oddim, guidance as og, lgp as olgp, unet as ounet
    from sketch2img_amd import synthetic
    from sketch2img_amd.config import SD15
    # 32 threads is the fastest setting on the GPU box's 256-thread host for this eager fp32 graph (measured
    # with tools/cpu_sweep.py: 5.3 s / UNet eval at 32 threads, 7.2 s at 64, 11.1 s at 128)
    cores = min(32, os.cpu_count() or 1)
    torch.set_num_threads(cores)
    cfg = ounet.SD15
    W = synthetic.unet_state_dict(SD15)
    sd = synthetic.lgp_state_dict(synthetic.lgp_input_dim(SD15))
    ehs = synthetic.text_embeddings(1)
    h, T = 32, 10
    x0, tgt = synthetic.initial_latents(0, 1, h), synthetic.sketch_targets(0, 1, h)
    runs = []
    for r in range(4):
        t0 = time.time()
        out = og.sample_one(cfg, W, dict(sd), ehs, x0, tgt, T)
        runs.append(time.time() - t0)
    assert torch.isfinite(out).all()
    # end-to-end parity number carried by every bench line: the same architecture / shape / seed,
    # UNGUIDED (a free-running guided trajectory separates chaotically), oracle here, HIP in main()
    t0 = time.time()
    with torch.no_grad():
        ref_unguided = og.sample_one(cfg, W, None, ehs, x0, None, T)
    t_unguided = time.time() - t0
    timed = runs[1:]
    mean = sum(timed) / len(timed)
    res = dict(value=1.0 / mean, unit="images/s", cores=cores, kind="port",
               sample=f"BASELINE config[0] as SURVEY 8(d) defines the CPU baseline: 1 sketch, 256x256, {T} DDIM steps, "
                      f"LGP guidance on steps 0..5, as-written formulation (autograd through both CFG rows), SD1.5 fp32 "
                      f"eager PyTorch on {cores} host threads; 1 warm-up ({runs[0]:.1f} s) + 3 timed runs "
                      f"({', '.join(f'{t:.1f}' for t in timed)} s), value = 1 / mean",
               runs_s=timed, warmup_s=runs[0], tflop_per_image=6.11,
               _parity=dict(x0=x0, ehs=ehs, ref=ref_unguided, T=T, seconds=t_unguided))
    if extrapolate_c2:
        h = 64
        x, tgt = synthetic.initial_latents(0, 1, h), synthetic.sketch_targets(0, 1, h)
        tab = oddim.make_tables(50)
        t = int(tab.timesteps[0])
        # the fp32 references of `config.eps_max`: full SD1.5, 2 CFG rows, 64 x 64 latents, at the first / middle / last timestep of the
        # schedule, the latents of samples 0 / 1 / 2 (the first evaluation doubles as the warm-up: first touch of the 64x64 buffers)
        eps_cases = []
        with torch.no_grad():
            for i, tt in enumerate((int(tab.timesteps[0]), int(tab.timesteps[len(tab.timesteps) // 2]), int(tab.timesteps[-1]))):
                xi = synthetic.initial_latents(i, 1, h)
                eps_cases.append(dict(x=xi, t=tt, ref=ounet.unet_forward(cfg, W, torch.cat([xi] * 2), tt, ehs)[0]))
        res["_parity"].update(eps_cases=eps_cases, eps_ehs=ehs)
        times = {}
        for guided in (True, False):
            t0 = time.time()
            x_in = torch.cat([x] * 2).requires_grad_(guided)
            with torch.enable_grad() if guided else torch.no_grad():
                eps, taps = ounet.unet_forward(cfg, W, x_in, t, ehs)
            eu, ec = eps.detach().chunk(2)
            nxt = oddim.ddim_step(tab, eu + 7.5 * (ec - eu), t, x)
            if guided:
                nxt = og.apply_anti_gradient(taps, sd, tab.alphas_cumprod, x_in, nxt, x, t, tgt, 1.6)
            times[guided] = time.time() - t0
        per_image = 26 * times[True] + 24 * times[False]
        res["config2_extrapolated"] = dict(
            value=1.0 / per_image, unit="images/s",
            sample=f"1 sample 512x512 after one warm-up evaluation: 1 guided step ({times[True]:.1f} s) + 1 unguided step "
                   f"({times[False]:.1f} s) measured, extrapolated to 26 guided + 24 unguided = {per_image:.0f} s/image")
    return res


# ---------------------------------------------------------------------------------------------------- workload
_SD_CACHE: dict = {}      # the (broadcast) synthetic state dicts: the accuracy-mode workload re-packs the same weights


def build_workload(args, rank, world, dev, dist, residual_fp32=None):
    """Everything resident in HBM: engines, inputs, tables.  Returns a dict with `one_batch()`.
    residual_fp32: override of the headline mode (the second timed region builds the other mode's twin of the workload)."""
    residual_fp32 = (args.mode == "tolerance") if residual_fp32 is None else residual_fp32
    from sketch2img_amd import synthetic
    from sketch2img_amd.config import SD15, SD21, SD_VAE, tap_channels
    from sketch2img_amd.dist import broadcast_state_dict, gather_images, gather_latents
    from sketch2img_amd.inject import HipInjector
    from sketch2img_amd.lgp import HipLGP
    from sketch2img_amd.sampler import DDIMTables, DPMTables, HipSampler
    from sketch2img_amd.unet import HipUNet
    from sketch2img_amd.vae import HipVAEDecoder

    C, T = args.config, args.ddim_steps
    cfg = SD21 if C == 5 else SD15
    h = 96 if C == 5 else 64
    S = args.samples_per_gpu or (4 if C == 5 else 8)
    first = args.first_sample + rank * S

    def weights(make, shapes=None, key=None):
        key = (key, C)
        if key not in _SD_CACHE:
            sd = make() if rank == 0 else None
            _SD_CACHE[key] = broadcast_state_dict(sd, shapes, dev, src=0) if dist is not None else sd
        sd = _SD_CACHE[key]
        return {k: v.clone() for k, v in sd.items()} if key[0] == "lgp" else sd      # (the LGP engine updates its BatchNorm statistics in place)

    sd_unet = weights(lambda: synthetic.unet_state_dict(cfg), synthetic.unet_param_shapes(cfg), "unet")
    guided = C == 2 and not args.no_guidance
    net = HipUNet(cfg, sd_unet, dev, need_backward=guided, residual_fp32=residual_fp32)
    lgp, target, sd_lgp = None, None, None
    if C == 2 and not guided:
        pass
    elif C == 2:
        sd_lgp = weights(lambda: synthetic.lgp_state_dict(synthetic.lgp_input_dim(cfg)), None, "lgp")
        lgp = HipLGP(sd_lgp, tap_channels(cfg), dev)
        target = synthetic.sketch_targets(first, S, h).to(dev)
    else:
        variant = "sketch" if C == 4 else "clip"
        sd_sat = weights(lambda: synthetic.satmixin_state_dict(cfg, variant), synthetic.satmixin_param_shapes(cfg, variant), "sat")
        inj = HipInjector(cfg, sd_sat, variant, dev)
        inj.set_scale(1.0)
        if C == 4:
            inj.set_res_samples(synthetic.res_samples(cfg, first, S, h))
        else:
            inj.set_state(synthetic.sketch_state(first, S))
        net.inject = inj
    vae = None
    if args.gather == "images":
        sd_vae = weights(lambda: synthetic.vae_decoder_state_dict(SD_VAE), synthetic.vae_decoder_param_shapes(SD_VAE), "vae")
        vae = HipVAEDecoder(SD_VAE, sd_vae, dev)
    ehs = synthetic.text_embeddings(S, dim=cfg.cross_attention_dim)
    net.prepare_context(ehs)
    tab = DPMTables.make(T) if args.scheduler == "dpm" else DDIMTables.make(T)
    net.prepare_timesteps(tab.timesteps.tolist())
    lat0 = synthetic.initial_latents(first, S, h).to(dev)
    sampler = HipSampler(net, lgp, use_graphs=args.graph)

    decode_events = []

    def one_batch():
        x = sampler.sample(lat0, target, T, tables=tab)
        if vae is not None:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            x = vae.decode_to_u8(x)                      # [S, 8h, 8h, 3] uint8: what the final gather carries
            e1.record()
            decode_events.append((e0, e1))
        if dist is not None:
            got = (gather_images if vae is not None else gather_latents)(x, world, dst=0)
            return x if got is None else torch.cat(got)
        return x

    return dict(one_batch=one_batch, sampler=sampler, net=net, decode_events=decode_events, vae=vae, S=S, h=h, T=T, tab=tab, lat0=lat0, target=target)


def _free_port() -> int:
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def self_launch(args) -> None:
    """`python bench.py --gpus N` (N > 1) outside a launcher: re-exec under torch.distributed.run, one rank per GPU, relay rank 0's
    JSON line and the exit code (the driver's scaling command must not die on a WORLD_SIZE assert)."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env["SKG_BENCH_SELF_LAUNCHED"] = "1"
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)      # (stderr passes through)
    lines = r.stdout.splitlines()
    js = [l for l in lines if l.startswith('{"metric"')]
    for l in lines:
        if l not in js[-1:]:
            print(l, file=sys.stderr)
    if js:
        print(js[-1], flush=True)
    sys.exit(r.returncode if (r.returncode or js) else 1)


# ------------------------------------------------------------------------------------------------ box calibration
def _smi_sample():
    """One `rocm-smi` reading: (sclk MHz, package power W, power cap W) - each None when the tool or the field is missing."""
    def run(*a):
        try:
            return subprocess.run(["rocm-smi", *a], capture_output=True, text=True, timeout=20).stdout
        except Exception:
            return ""
    dev = os.environ.get("SKG_BENCH_SMI_DEVICE")          # (HIP_VISIBLE_DEVICES may renumber: default = the first card rocm-smi lists)
    sel = ["-d", dev] if dev else []
    out = run(*sel, "--showpower", "--showclocks")
    cap = run(*sel, "--showmaxpower")
    ck = re.search(r"sclk[^(\n]*\((\d+)Mhz\)", out)
    pw = re.search(r"Power[^:\n]*\(W\)\s*:\s*([0-9.]+)", out) or re.search(r"Power[^:\n]*:\s*([0-9.]+)", out)
    cp = re.search(r"Max[^:\n]*Power[^:\n]*:\s*([0-9.]+)", cap) or re.search(r"Power[^:\n]*:\s*([0-9.]+)", cap)
    return (int(ck.group(1)) if ck else None, float(pw.group(1)) if pw else None, float(cp.group(1)) if cp else None)


def box_probe(dev, seconds: float = 1.6):
    """What THIS box sustains (the same kernels read 11-12 % apart on two driver boxes): the in-library bare
    MFMA stream (skg_box_probe_mfma: register-resident 16x16x32 fp16 MFMAs on pseudo-random operands, 512 workgroups x 8 waves)
    runs for `seconds`; the LAST third is timed with HIP events (the clock has settled under the power limit by then) while
    one rocm-smi reading is taken beside it.  -> flat scalars for the contract line."""
    from sketch2img_amd._lib import check, lib
    out = torch.empty(512 * 512, device=dev, dtype=torch.float32)
    st = torch.cuda.current_stream().cuda_stream
    iters = 4096
    flop = 512 * 8 * iters * 40 * 2 * 16 * 16 * 32

    def burst(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            check(lib.skg_box_probe_mfma(out.data_ptr(), iters, st), "skg_box_probe_mfma")
        e1.record()
        return e0, e1

    e0, e1 = burst(4)
    torch.cuda.synchronize()
    per = e0.elapsed_time(e1) * 1e-3 / 4                       # seconds per launch (first estimate)
    n = max(6, int(seconds / max(per, 1e-4)))
    smi = {}
    th = threading.Thread(target=lambda: smi.update(v=_smi_sample()))
    a0, a1 = burst(n - n // 3)                                 # queued asynchronously: the GPU is busy for ~seconds from here
    b0, b1 = burst(n // 3)
    th.start()                                                 # ... and the reading is taken while it is
    th.join()
    torch.cuda.synchronize()
    sclk, power, cap = smi.get("v", (None, None, None))
    t_tail = b0.elapsed_time(b1) * 1e-3
    return dict(box_mfma_tflops=flop * (n // 3) / t_tail / 1e12, box_sclk_mhz=sclk, box_power_w=power, box_power_cap_w=cap,
                box_probe_s=(a0.elapsed_time(b1)) * 1e-3)


def plumbing_check(args, world, rank):
    """Tests only (no GPU needed): everything of the N > 1 launch path except the hot path - rendezvous, barrier, max over ranks,
    rank 0's single JSON line - with the gloo backend."""
    import torch.distributed as dist
    dist.init_process_group("gloo")
    t0 = time.perf_counter()
    dist.barrier()
    tt = torch.tensor([time.perf_counter() - t0 + rank], dtype=torch.float64)
    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
    ranks = [None] * world
    dist.all_gather_object(ranks, (rank, int(os.environ.get("LOCAL_RANK", "-1"))))
    if rank == 0:
        print(json.dumps({"metric": "plumbing-check (no hot path)", "value": 0.0, "unit": "images/s", "n_gpus": world, "steps": args.steps,
                          "warmup": args.warmup, "ranks": ranks, "max_over_ranks": float(tt),
                          "self_launched": os.environ.get("SKG_BENCH_SELF_LAUNCHED") == "1"}), flush=True)
    dist.destroy_process_group()


def _timed(one_batch, steps, warmup, barrier, dist, dev):
    out = None
    for _ in range(warmup):
        out = one_batch()
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = one_batch()
    torch.cuda.synchronize(); barrier(); torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt)
    return dt, out


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        self_launch(args)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    assert world == args.gpus, f"--gpus {args.gpus} but WORLD_SIZE={world} (an external launcher started another number of ranks)"
    if args.plumbing_check:
        return plumbing_check(args, world, rank)
    # test-only switches: run the N > 1 code path with several ranks on ONE GPU (gloo carries CUDA tensors)
    backend = os.environ.get("SKG_BENCH_BACKEND", "nccl")
    if "SKG_BENCH_DEVICE" in os.environ:
        local = int(os.environ["SKG_BENCH_DEVICE"])
    assert torch.cuda.is_available() and local < torch.cuda.device_count(), \
        (f"rank {rank}: no cuda:{local} ({torch.cuda.device_count()} visible device(s)); several ranks on one device is a test "
         "configuration: SKG_BENCH_DEVICE=0 SKG_BENCH_BACKEND=gloo")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    # SKG_BENCH_FORCE_DIST=1: initialise the process group even at world size 1, so that the weight broadcast and the
    # image gather issue real RCCL calls on device tensors on the one GPU a test box has (tests/test_gpu_configs.py)
    if world > 1 or os.environ.get("SKG_BENCH_FORCE_DIST") == "1":
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)    # RCCL over xGMI
        else:
            dist.init_process_group(backend)

    from sketch2img_amd import ops

    tol = args.mode == "tolerance"
    global _PROFILE_TAG
    _PROFILE_TAG = "" if tol else "fast"
    box = {}
    if rank == 0 and not args.no_box_probe:
        box = box_probe(dev)

    t_setup = time.time()
    wl = build_workload(args, rank, world, dev, dist)
    one_batch, S, T, C = wl["one_batch"], wl["S"], wl["T"], args.config
    torch.cuda.synchronize()
    t_setup = time.time() - t_setup
    if dist is not None:      # N ranks build their weight packs beside each other on one host: report the slowest
        ts = torch.tensor([t_setup], device=dev, dtype=torch.float64)
        dist.all_reduce(ts, op=dist.ReduceOp.MAX)
        t_setup = float(ts)

    def barrier():
        if dist is not None:
            dist.barrier()

    for _ in range(args.warmup):
        one_batch()
    wl["decode_events"].clear()
    dt, out = _timed(one_batch, args.steps, 0, barrier, dist, dev)
    lat_final = wl["sampler"].last_latents
    finite = bool(torch.isfinite(lat_final).all())
    if dist is not None:
        ft = torch.tensor([int(finite)], device=dev)
        dist.all_reduce(ft, op=dist.ReduceOp.MIN)
        finite = bool(int(ft))
    value = world * S * args.steps / dt
    # the VAE decode's share of the timed region on this rank (HIP events), and the throughput with round 1's definition
    # of a step (stops at the latents: SURVEY 8d keeps the 2.51 TFLOP decode out of F_img)
    decode_s = sum(a.elapsed_time(b) for a, b in wl["decode_events"]) * 1e-3
    if rank == 0 and args.dump_images:
        torch.save(dict(images=out.cpu(), latents=lat_final.cpu()), args.dump_images)

    roof, cpu = None, None
    if rank == 0 and not args.no_roofline:
        fork, wl["sampler"].fork_guidance = wl["sampler"].fork_guidance, False      # every launch timed ALONE (no second stream)
        with LaunchTimer(ops) as lt:
            wl["sampler"].sample(wl["lat0"], wl["target"], T, tables=wl["tab"], graphs=False)
            if wl["vae"] is not None:
                wl["vae"].decode_to_u8(wl["sampler"].last_latents)
        wl["sampler"].fork_guidance = fork
        agg = lt.summary()
        if args.shape_report:
            lt.shape_report(args.shape_report)
        name, (n, fl, sec, nb, _, _) = max(agg.items(), key=lambda kv: kv[1][2])
        tot_sec = sum(v[2] for v in agg.values())
        traffic, traffic_file = pmc_traffic(C, name)
        byop = by_operator(agg)
        roof = dict(bound="mfma", kernel=name, achieved=fl / sec / 1e12, peak=PEAK_FP16_TFLOPS, unit="TFLOP/s",
                    frac=fl / sec / 1e12 / PEAK_FP16_TFLOPS, traffic=traffic, algorithmic_bytes=nb / n,
                    traffic_source=(f"committed {traffic_file}: rocprofv3 --pmc passes, (2*FETCH_SIZE + WRITE_SIZE) KiB per launch") if traffic_file else None,
                    launches=n, avg_launch_us=sec / n * 1e6, avg_launch_gflop=fl / n / 1e9,
                    timing="HIP events on the launch stream around every launch of one instrumented batch, guidance branch in line",
                    all_contraction_tflops=sum(v[1] for v in agg.values()) / tot_sec / 1e12,
                    contraction_share_of_step=tot_sec / (dt / args.steps),
                    # every launch against ITS OWN bound, max(flops / 2.5 PFLOP/s, algorithmic bytes / 8 TB/s): the short-K
                    # projections (K = 320: 320 flop per output byte against a machine balance of 312) are HBM-bound launches
                    # of the same instantiation that runs the MFMA-bound ones
                    all_contraction_frac_of_own_roofline=sum(v[4] for v in agg.values()) / tot_sec)
        # FLAT per-operator fractions of the MFMA peak (an operator may run on several instantiations: the 3x3 convolution on
        # gemm2_kernel<..., 1, ...> and, for the 64x64 level, on gemm8_kernel<320, 1, ...>)
        for op_name, key in (("conv3x3", "conv3x3_frac"), ("gemm", "gemm_frac"), ("attention_fwd", "attn_fwd_frac"), ("ff_block", "ff_block_frac"),
                             ("xattn_block", "xattn_block_frac"), ("conv3x3_resample", "conv_resample_frac")):
            roof[key] = byop[op_name]["frac"] if op_name in byop else None
        rp = rocprof_duration(C, name)
        roof["rocprof_frac"] = roof["rocprof_avg_launch_us"] = roof["rocprof_source"] = None
        if rp is not None:
            avg_ns, calls, f = rp
            roof["rocprof_frac"] = fl / n / avg_ns / 1e3 / PEAK_FP16_TFLOPS
            roof["rocprof_avg_launch_us"] = avg_ns / 1e3
            roof["rocprof_source"] = f"committed {f} (rocprofv3 --kernel-trace --stats of this command), {calls} launches"
        # the nested detail (parsers that keep scalars only drop these; everything needed to judge the line is flat above)
        roof["by_operator"] = byop
        roof["per_kernel"] = {k: dict(launches=v[0], tflops=v[1] / v[2] / 1e12, seconds=v[2],
                                      frac_of_own_roofline=v[4] / v[2], hbm_bound_share_of_time=v[5] / v[2])
                              for k, v in sorted(agg.items(), key=lambda kv: -kv[1][2])}
    # ---- second timed region: the SAME workload in the OTHER mode (headline = accuracy mode -> the all-fp16 mode here: the reference's
    # own GPU configuration, app.py:34, which misses north_star's eps bound; --fast-fp16 swaps the two)
    second, wl2 = None, None
    want2 = not args.no_second_mode and not args.graph and args.scheduler == "ddim" and not args.no_guidance
    if want2 and world > 1 and not args.second_mode_multi:
        # (the scaling runs: a second full workload per rank - weights, packs, broadcast - would only lengthen them; the mode's
        # cost is a per-GPU figure and is measured at N = 1)
        second, want2 = dict(skipped="n_gpus > 1: measured at N = 1 (pass --second-mode-multi to time it here)"), False
    if want2:
        wl2 = build_workload(args, rank, world, dev, dist, residual_fp32=not tol)
        dt2, _ = _timed(wl2["one_batch"], args.second_mode_steps, 1, barrier, dist, dev)
        fin2 = bool(torch.isfinite(wl2["sampler"].last_latents).all())
        if dist is not None:
            tt = torch.tensor([float(not fin2)], device=dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            fin2 = not bool(tt[0])
        v2 = world * S * args.second_mode_steps / dt2
        second = dict(value=v2, unit="images/s", ms_per_step=dt2 / args.second_mode_steps * 1e3, steps=args.second_mode_steps, warmup=1,
                      outputs_finite=fin2, mode="fast_fp16" if tol else "residual_fp32")
        finite = finite and fin2
    eps = {}
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()
        par = cpu.pop("_parity")
        if C == 2 and "eps_cases" in par and S >= len(par["eps_cases"]) and args.first_sample == 0:
            # eps of full-size evaluations (2 CFG rows, 64 x 64 latents; first / middle / last timestep, three samples) of the mode(s)
            # built above against the fp32 CPU oracle
            from sketch2img_amd.unet import CIN_PAD
            nets = {("residual_fp32" if tol else "fast_fp16"): wl["net"]}
            if wl2 is not None:
                nets["fast_fp16" if tol else "residual_fp32"] = wl2["net"]
            # ... evaluated AT THE WORKLOAD'S OWN BATCH (all S samples = 2 S rows in one evaluation, the sampler's call form): the kernel
            # instantiations a launch takes depend on the batch size (256 x 320 tiles, split factors, the Winograd path of the small maps), so a
            # 2-row evaluation would measure other kernels than the timed region ran; rows (i, S + i) = sample i against the oracle's 2-row
            # evaluation of that sample alone (samples are independent)
            h_ = wl["h"]
            x_all = torch.cat([wl["lat0"], wl["lat0"]]).to(dev, torch.float32).contiguous()
            x32 = ops.nchw_to_nhwc(x_all, CIN_PAD)
            for mode_name, net in nets.items():
                mx = rel = std = 0.0
                for i, case in enumerate(par["eps_cases"]):
                    assert i < S and torch.equal(case["x"], wl["lat0"][i:i + 1].cpu())
                    e, _ = net.forward(x32, case["t"], 2 * S, h_, want_taps=False, shared_input=True)
                    got = ops.nhwc_to_nchw(e, 2 * S, 4, h_, h_).cpu()
                    d = torch.stack([got[i], got[S + i]]) - case["ref"]
                    mx, rel = max(mx, float(d.abs().max())), max(rel, float(d.norm() / case["ref"].norm()))
                    std = max(std, float(case["ref"].std()))
                eps[mode_name] = dict(eps_max=mx, eps_rel=rel, eps_std=std, evals=len(par["eps_cases"]), rows_per_eval=2 * S)
        if C == 2:
            # the HIP path on the oracle's inputs: full SD1.5, 1 sample, 32 x 32 latents, 10 unguided DDIM steps, free running
            from sketch2img_amd.sampler import HipSampler
            net = wl["net"]
            saved_ctx = net.ctx
            net.prepare_context(par["ehs"])
            got = HipSampler(net, None).sample(par["x0"], None, par["T"]).cpu()
            net.ctx = saved_ctx
            d = got - par["ref"]
            cpu["hip_vs_oracle_rel"] = float(d.norm() / par["ref"].norm())
            cpu["hip_vs_oracle_max_abs"] = float(d.abs().max())
            cpu["hip_vs_oracle"] = ("end latents of the HIP path vs the fp32 CPU oracle on identical inputs: full SD1.5, 1 sample, "
                                    f"256x256, {par['T']} UNGUIDED DDIM steps, CFG 7.5, free running (oracle run: {par['seconds']:.1f} s); "
                                    "relative Frobenius / max abs")

    if rank == 0:
        sched = "DPM-Solver++ 2M" if args.scheduler == "dpm" else "DDIM"
        metric = "sketch-guided images/sec whole-node, SD1.5 512px 50-step DDIM"
        if C == 5:
            metric = "sketch-guided images/sec whole-node, SD2.1 768px 50-step DDIM (clip_guided_attn)"
        mode_name = "residual_fp32" if tol else "fast_fp16"
        fmt = dict(S=S, T=T, G=int(0.5 * T), sched=sched)
        variant = (" - INFORMATIONAL VARIANT: guidance off" if args.no_guidance else "")
        mine = eps.get(mode_name, {})
        other = eps.get("fast_fp16" if tol else "residual_fp32", {})
        config = {
            "workload": WORKLOADS[C].format(**fmt) + (" (guidance OFF)" if args.no_guidance else ""),
            "baseline_config": C, "samples_per_gpu": S, "global_batch": world * S, "ddim_steps": T,
            "scheduler": args.scheduler, "hip_graphs": bool(args.graph),
            # ---- the mode `value` was timed in, and its accuracy against the fp32 CPU oracle (flat scalars: VERDICT r5 next #1)
            "mode": mode_name,                       # residual_fp32 = the north_star-compliant accuracy mode
            "eps_bound": EPS_BOUND,
            "eps_max": mine.get("eps_max"),         # worst max |eps - eps_fp32 oracle| over `eps_evals` full-size evaluations (this run)
            "eps_rel": mine.get("eps_rel"),         # worst relative Frobenius distance (scale-free)
            "eps_evals": mine.get("evals"), "eps_rows_per_eval": mine.get("rows_per_eval"),
            # the same absolute error for a UNIT-VARIANCE eps (a trained checkpoint): the synthetic model's eps has std ~0.37 and the
            # error scales with conv_out (tests/test_gpu_configs.py checks the power-of-two rescale) - reported, not asserted
            "eps_max_unit_var": (mine["eps_max"] / mine["eps_std"]) if mine else None,
            "eps_std": mine.get("eps_std"),
            # ---- the other mode, timed beside it (three batches after one warm-up)
            ("fast_fp16_value" if tol else "residual_fp32_value"): second.get("value") if second else None,
            ("fast_fp16_ms_per_step" if tol else "residual_fp32_ms_per_step"): second.get("ms_per_step") if second else None,
            ("fast_fp16_eps_max" if tol else "residual_fp32_eps_max"): other.get("eps_max"),
            "mode_cost": (1.0 - (value / second["value"] if tol else second["value"] / value)) if second and "value" in second else None,
            # ---- box calibration: what a bare MFMA stream sustains on THIS box, and one rocm-smi reading under it
            "box_mfma_tflops": box.get("box_mfma_tflops"), "box_sclk_mhz": box.get("box_sclk_mhz"),
            "box_power_w": box.get("box_power_w"), "box_power_cap_w": box.get("box_power_cap_w"),
            "value_per_box_pflops": (value / world / (box["box_mfma_tflops"] / 1e3)) if box.get("box_mfma_tflops") else None,
            "hip_streams": (2 if (C == 2 and not args.no_guidance and wl["sampler"].fork_guidance and not args.graph) else 1),
            "output": ("decoded uint8 images [S, H, W, 3] (VAE decode on-rank, inside the timed region"
                       + (", gathered on rank 0)" if world > 1 else ")")) if args.gather == "images"
            else "fp32 latents (no decode)",
            "parallelism": (f"replicas x{world} (samples sharded, weights broadcast, "
                            f"{'decoded images' if args.gather == 'images' else 'latents'} gathered"
                            + (f"; backend {dist.get_backend()}" + (" = RCCL" if dist.get_backend() == "nccl" else "") + ")"
                               if dist is not None else "; single process, no process group)")),
        }
        res = {
            "metric": metric,
            "value": value, "unit": "images/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "fp16", "data": "synthetic",
            "config": config,
            "workload_detail": WORKLOAD_DETAILS[C].format(**fmt) + variant
            + (", UNet in its accuracy mode (residual stream and the conv outputs that feed a norm as (hi, lo) fp16 pairs: "
               "AntiGradientPipeline.from_pretrained(..., residual_fp32=True))" if tol else ", every stored tensor fp16"),
            "ms_per_image": dt / args.steps / S * 1e3,
            "decode_ms_per_step": decode_s / args.steps * 1e3 if wl["decode_events"] else None,
            "value_excluding_decode": world * S * args.steps / (dt - decode_s) if wl["decode_events"] and world == 1 else None,
            "achieved_tflops_per_gpu": value / world * f_img_tflop(C, T) if args.scheduler == "ddim" and not args.no_guidance else None,
            "tflop_per_image": f_img_tflop(C, T), "outputs_finite": finite, "out_shape": list(out.shape),
            "setup_s": t_setup, "second_mode": second, "eps_vs_fp32_oracle": eps or None, "box": box or None,
            "roofline": roof, "cpu_baseline": cpu,
        }
        print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()
    if not finite:
        sys.exit("bench.py: non-finite latents - the throughput above is INVALID")


if __name__ == "__main__":
    main()

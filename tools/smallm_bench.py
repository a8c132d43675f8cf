#!/usr/bin/env python3
"""The small-M bucket: every GEMM / 3x3 convolution of a config-2 batch that launches at most 256
tiles of 128 x 160 - the 16x16 / 8x8 levels and the cond-only backward - timed on the k-pair kernel (gemmk.hip, default)
and on gemm2.hip (SKG_GEMMK=0: one 4-wave workgroup per CU or split-K + reduce), interleaved rounds in ONE process
(the switch is read at every launch).  Weights rotate through a pool larger than the 256 MB Infinity Cache (in the UNet
every layer's weights come from HBM: 1.7 GB are streamed per evaluation), activations stay warm (their producer just wrote
them).  `n` = launches of that shape in one config-2 batch (profiles/r02_cfg2_shapes.txt) -> weighted totals.

    python tools/smallm_bench.py [--rounds 5] [--iters 20] [--out FILE]
    SKG_LIB=sketch2img_amd/libskg_lab.so python tools/smallm_bench.py --probes      # + SKG_GK_EXP ceiling probes (make lab)
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import ops  # noqa: E402
from sketch2img_amd._lib import lib  # noqa: E402

DEV = "cuda:0"
GEMMS = [  # M, N, K, residual, launches per batch
    (4096, 1280, 1280, True, 750), (4096, 1280, 1280, False, 500), (2048, 1280, 1280, False, 650), (8192, 640, 640, False, 650),
    (4096, 1280, 5120, True, 250), (2048, 1280, 10240, False, 130), (8192, 640, 5120, False, 130), (2048, 1280, 3840, False, 130),
    (1024, 1280, 2560, False, 150), (8192, 640, 1920, False, 130), (4096, 1280, 2560, False, 100), (1024, 1280, 1280, True, 150),
    (512, 1280, 1280, False, 130), (1024, 1280, 1280, False, 100), (512, 2560, 1280, False, 78), (1024, 1280, 5120, True, 50),
    (4096, 1280, 1920, False, 50), (2048, 2560, 1280, False, 52), (512, 1280, 10240, False, 26), (4096, 1280, 640, False, 50),
    (512, 1280, 3840, False, 26), (2048, 640, 1280, False, 26),
]
CONVS = [  # rows, hw, cin, cout, residual, launches per batch
    (16, 16, 1280, 1280, True, 250), (16, 16, 2560, 1280, False, 100), (16, 8, 1280, 1280, True, 350), (8, 16, 1280, 1280, False, 182),
    (8, 8, 1280, 1280, False, 286), (16, 8, 2560, 1280, False, 150), (16, 8, 1280, 1280, False, 200), (16, 16, 1920, 1280, False, 50),
    (8, 16, 1280, 2560, False, 52), (16, 16, 1280, 1280, False, 50), (8, 8, 1280, 2560, False, 78), (8, 32, 640, 640, False, 156),
    (8, 16, 1280, 1920, False, 26), (16, 16, 640, 1280, False, 50), (8, 16, 1280, 640, False, 26), (8, 32, 640, 320, False, 26),
]
POOL_BYTES = 640 << 20


def time_variants(run, variants, rounds, iters):
    """run(i): one launch using weight copy i.  Returns {variant: best-of-rounds mean us}."""
    best = {v: float("inf") for v in variants}
    for _ in range(rounds):
        for v, env in variants.items():
            for k, val in env.items():
                os.environ[k] = val
            for i in range(3):
                run(i)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(iters):
                run(i)
            e1.record()
            torch.cuda.synchronize()
            best[v] = min(best[v], e0.elapsed_time(e1) * 1e3 / iters)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--iters", type=int, default=20)
    ap.add_argument("--probes", action="store_true", help="lab build: + no-DMA / no-store / both probes of the k-pair kernel")
    ap.add_argument("--splits", action="store_true", help="gemm2.hip only: cap the cross-workgroup K slices at 8 (ships) / 4 / 2")
    ap.add_argument("--plain", action="store_true", help="the shipped kernels only (no lab variants)")
    ap.add_argument("--stages", action="store_true", help="lab build, gemm2.hip only: three LDS stages (ships) against four / six (SKG_NS)")
    ap.add_argument("--split-stages", action="store_true", help="gemm2.hip only: split-K launches on 2 stages x 512 workgroups (ships before round 5) "
                    "against 3 / 4 stages x 256 workgroups (SKG_SPLIT_NS / SKG_SPLIT_TARGET)")
    ap.add_argument("--pool-mb", type=int, default=640, help="weight pool per shape; 0: two copies only = weights warm in the Infinity Cache")
    ap.add_argument("--out", default=None)
    args = ap.parse_args()
    global POOL_BYTES
    POOL_BYTES = args.pool_mb << 20
    variants = {"gemm2": {"SKG_GEMMK": "0", "SKG_GK_EXP": "0"}, "kpair": {"SKG_GEMMK": "1", "SKG_GK_EXP": "0"}}
    if args.probes:
        variants.update({"kpair-noDMA": {"SKG_GEMMK": "1", "SKG_GK_EXP": "1"}, "kpair-noStore": {"SKG_GEMMK": "1", "SKG_GK_EXP": "2"},
                         "kpair-neither": {"SKG_GEMMK": "1", "SKG_GK_EXP": "3"}})
    if args.plain:
        variants = {"shipped": {}}
    if args.stages:
        variants = {"ns3": {"SKG_GEMMK": "0", "SKG_GK_EXP": "0", "SKG_NS": "0"}, "ns4": {"SKG_GEMMK": "0", "SKG_GK_EXP": "0", "SKG_NS": "4"},
                    "ns6": {"SKG_GEMMK": "0", "SKG_GK_EXP": "0", "SKG_NS": "6"}}
    if args.splits:
        variants = {f"max{n}": {"SKG_GEMMK": "0", "SKG_GK_EXP": "0", "SKG_MAX_SPLITS": str(n)} for n in (8, 4, 2)}
    if args.split_stages:
        base = {"SKG_GEMMK": "0", "SKG_GK_EXP": "0"}
        variants = {"ns2x512": dict(base, SKG_SPLIT_NS="2", SKG_SPLIT_TARGET="512"), "ns3x512": dict(base, SKG_SPLIT_NS="3", SKG_SPLIT_TARGET="512"),
                    "ns3x256": dict(base, SKG_SPLIT_NS="3", SKG_SPLIT_TARGET="256"), "ns4x256": dict(base, SKG_SPLIT_NS="4", SKG_SPLIT_TARGET="256"),
                    "ns4x512": dict(base, SKG_SPLIT_NS="4", SKG_SPLIT_TARGET="512")}
    g = torch.Generator().manual_seed(1)
    lines, tot = [], {v: {"gemm": 0.0, "conv": 0.0} for v in variants}
    flops = {"gemm": 0.0, "conv": 0.0}
    hdr = f"{'shape':44s} {'n':>4s} " + " ".join(f"{v:>14s}" for v in variants) + "   TF/s: " + " ".join(f"{v:>8s}" for v in variants)
    lines.append(hdr)
    for M, N, K, res, n in GEMMS:
        os.environ["SKG_GEMMK"] = "1"
        assert args.splits or args.plain or args.stages or args.split_stages or lib.skg_gemm_variant(M, N, K, 0, 0) == 9160, (M, N, K)
        copies = max(2, min(64, POOL_BYTES // (N * K * 2)))
        W = [(torch.randn(N, K, generator=g) * K ** -0.5).half().to(DEV) for _ in range(2)]
        W = W + [W[i % 2].clone() for i in range(copies - 2)]
        a = torch.randn(M, K, generator=g).half().to(DEV)
        b = torch.randn(N, generator=g).half().to(DEV)
        r = torch.randn(M, N, generator=g).half().to(DEV) if res else None
        out = torch.empty(M, N, device=DEV, dtype=torch.float16)
        t = time_variants(lambda i: ops.gemm(a, W[i % copies], out=out, bias=b, residual=r), variants, args.rounds, args.iters)
        fl = 2.0 * M * N * K
        flops["gemm"] += fl * n
        for v in variants:
            tot[v]["gemm"] += t[v] * n
        lines.append(f"{f'gemm M{M} N{N} K{K}' + ('+res' if res else ''):44s} {n:4d} " + " ".join(f"{t[v]:11.1f} us" for v in variants)
                     + "         " + " ".join(f"{fl / t[v] / 1e6:8.0f}" for v in variants))
        print(lines[-1], flush=True)
        del W
    for rows, hw, cin, cout, res, n in CONVS:
        M = rows * hw * hw
        os.environ["SKG_GEMMK"] = "1"
        assert args.splits or args.plain or args.stages or args.split_stages or lib.skg_gemm_variant(M, cout, 9 * cin, cin, 1) == 9160, (rows, hw, cin, cout)
        copies = max(2, min(64, POOL_BYTES // (cout * 9 * cin * 2)))
        W = [(torch.randn(cout, 9 * cin, generator=g) * (9 * cin) ** -0.5).half().to(DEV) for _ in range(2)]
        W = W + [W[i % 2].clone() for i in range(copies - 2)]
        x = torch.randn(M, cin, generator=g).half().to(DEV)
        b = torch.randn(cout, generator=g).half().to(DEV)
        r = torch.randn(M, cout, generator=g).half().to(DEV) if res else None
        out = torch.empty(M, cout, device=DEV, dtype=torch.float16)
        t = time_variants(lambda i: ops.conv3x3(x, W[i % copies], rows, hw, hw, 0, out=out, bias=b, residual=r), variants,
                          args.rounds, args.iters)
        fl = 2.0 * M * cout * 9 * cin
        flops["conv"] += fl * n
        for v in variants:
            tot[v]["conv"] += t[v] * n
        lines.append(f"{f'conv S1 M{M} Cin{cin} Cout{cout}' + ('+res' if res else ''):44s} {n:4d} " + " ".join(f"{t[v]:11.1f} us" for v in variants)
                     + "         " + " ".join(f"{fl / t[v] / 1e6:8.0f}" for v in variants))
        print(lines[-1], flush=True)
        del W
    for kind in ("gemm", "conv"):
        lines.append(f"{kind} bucket, launches-per-batch weighted: " + "   ".join(
            f"{v} {tot[v][kind] * 1e-3:.1f} ms = {flops[kind] / tot[v][kind] / 1e6:.0f} TF/s" for v in variants))
        print(lines[-1])
    if args.out:
        open(args.out, "w").write("\n".join(lines) + "\n")


if __name__ == "__main__":
    main()

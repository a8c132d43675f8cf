#!/usr/bin/env python3
"""A/B of the forward-attention variants (SKG_ATTN_VAR, attention.hip: 0 = shipped, 7 = two register prefetch sets at d = 64; the
round-2 sweep over five more variants is recorded in profiles/r02_attn_variants.txt) on the config-2 / config-5 shapes: one
subprocess per variant (the variant is read once per process), interleaved rounds, correctness of every variant checked
against an fp32 torch reference on two (row, head) pairs.   python tools/attn_var_bench.py [variants...]"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHAPES = [(16, 8, 4096, 4096, 40), (8, 5, 9480, 9473, 64), (8, 10, 2568, 2561, 64), (8, 20, 840, 833, 64),
          (16, 8, 1024, 1024, 80)]


def worker():
    import torch
    sys.path.insert(0, ROOT)
    from sketch2img_amd import ops
    dev = "cuda:0"
    for B, H, Nq, Nkv, d in SHAPES:
        C = H * d
        kvs = (Nkv + 7) // 8 * 8
        g = torch.Generator(device="cpu").manual_seed(5)
        q = torch.randn(B * Nq, C, generator=g).half().to(dev)
        k = torch.randn(B * kvs, C, generator=g).half().to(dev)
        v = torch.randn(B * kvs, C, generator=g).half().to(dev)
        vt = ops.transpose(v)
        sc = d ** -0.5
        o = ops.attn_fwd(q, k, vt, B, H, Nq, Nkv, kvs, d, sc)
        err = 0.0
        for b, h in ((0, 0), (B - 1, H - 1)):
            qq = q[b * Nq:(b + 1) * Nq, h * d:(h + 1) * d].float()
            kk = k[b * kvs:b * kvs + Nkv, h * d:(h + 1) * d].float()
            vv = v[b * kvs:b * kvs + Nkv, h * d:(h + 1) * d].float()
            ref = torch.softmax(qq @ kk.t() * sc, -1) @ vv
            got = o[b * Nq:(b + 1) * Nq, h * d:(h + 1) * d].float()
            err = max(err, float((got - ref).norm() / ref.norm()))
        ts = []
        for _ in range(3):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                ops.attn_fwd(q, k, vt, B, H, Nq, Nkv, kvs, d, sc)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / 10 * 1e3)
        t = min(ts)
        fl = 4.0 * B * H * Nq * Nkv * d
        print(f"VAR {os.environ.get('SKG_ATTN_VAR', '0')} B{B} H{H} Nq{Nq} Nkv{Nkv} d{d}: {t:8.1f} us {fl / t / 1e6:7.1f} TF/s  rel err {err:.2e}"
              + ("  WRONG" if err > 2e-3 else ""), flush=True)


if __name__ == "__main__":
    if os.environ.get("SKG_ATTN_WORKER"):
        worker()
    else:
        for v in (sys.argv[1:] or ["0", "7", "0", "7"]):
            r = subprocess.run([sys.executable, __file__], env=dict(os.environ, SKG_ATTN_VAR=v, SKG_ATTN_WORKER="1"),
                               capture_output=True, text=True)
            print(r.stdout, r.stderr[-800:] if r.returncode else "", flush=True)

"""A/B of the fused feed-forward sub-block (skg_ff_block_f16) against the three launches it replaces, at the shapes of a
config-2 / config-5 batch: LayerNorm -> FF1 (+ fused gate) -> FF2 + residual, C = 320, F = 1280.

  python tools/ff_block_bench.py [--rows 65536 32768 73728] [--reps 20] [--pool 12]

Inside a UNet evaluation neither the activations nor the weights of a layer are in L2 when it starts, so every repetition
works on another of `--pool` (x, weights) sets (12 sets x (42 MB x + 2.4 MB pack) >> the 32 MB of L2; the 256 MB
Infinity Cache still helps both sides equally).  Prints per-chain HIP-event times and the fused kernel's TFLOP/s."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from sketch2img_amd import ops
from sketch2img_amd.unet import pack_ff_block


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="+", default=[65536, 32768, 73728])
    ap.add_argument("--reps", type=int, default=24)
    ap.add_argument("--pool", type=int, default=12)
    a = ap.parse_args()
    d = torch.device("cuda:0")
    C, Fh = 320, 1280
    g = torch.Generator().manual_seed(0)
    sets = []
    for i in range(a.pool):
        w1 = (torch.randn(2 * Fh, C, generator=g) * C ** -0.5).half()
        b1 = (torch.randn(2 * Fh, generator=g) * 0.1).half()
        w2 = (torch.randn(C, Fh, generator=g) * Fh ** -0.5).half()
        b2 = (torch.randn(C, generator=g) * 0.1).half()
        idx = ops.geglu_interleave_index(Fh)
        pack, bias1 = pack_ff_block(w1, b1, w2, d)
        sets.append(dict(pack=pack, bias1=bias1, b2=b2.to(d), w1i=w1[idx].contiguous().to(d), b1i=b1[idx].contiguous().to(d),
                         w2=w2.to(d), gam=torch.ones(C, device=d, dtype=torch.float16), bet=torch.zeros(C, device=d, dtype=torch.float16)))
    for M in a.rows:
        xs = [torch.randn(M, C, device=d, dtype=torch.float16) for _ in range(a.pool)]
        out = torch.empty(M, C, device=d, dtype=torch.float16)
        gg = torch.empty(M, Fh, device=d, dtype=torch.float16)
        a3 = torch.empty(M, C, device=d, dtype=torch.float16)

        def fused(i):
            s = sets[i % a.pool]
            ops.ff_block(xs[i % a.pool], s["gam"], s["bet"], 1e-5, s["pack"], s["bias1"], s["b2"], out=out)

        def three(i):
            s = sets[i % a.pool]
            x = xs[i % a.pool]
            ops.layernorm(x, s["gam"], s["bet"], 1e-5, out=a3)
            ops.gemm(a3, s["w1i"], bias=s["b1i"], geglu=True, out=gg)
            ops.gemm(gg, s["w2"], out, bias=s["b2"], residual=x)

        res = {}
        for name, fn in (("three launches", three), ("fused", fused), ("three launches", three), ("fused", fused)):
            for i in range(3):
                fn(i)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(a.reps):
                fn(i)
            e1.record()
            torch.cuda.synchronize()
            res.setdefault(name, []).append(e0.elapsed_time(e1) / a.reps * 1e3)
        flops = 2.0 * M * C * 2 * Fh + 2.0 * M * Fh * C
        t3, tf = min(res["three launches"]), min(res["fused"])
        print(f"M {M:6d}: three launches {t3:7.1f} us   fused {tf:7.1f} us ({flops / tf * 1e-6:6.0f} TFLOP/s)   x{t3 / tf:.2f}   "
              f"runs {['%.1f' % v for v in res['three launches']]} / {['%.1f' % v for v in res['fused']]}", flush=True)


if __name__ == "__main__":
    main()

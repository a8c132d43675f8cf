"""A/B of the fused cross-attention sub-block (skg_xattn_block_f16) against the four launches it replaces at the 64 x 64
level of SD1.5: LayerNorm -> to_q -> attention over the 77 text keys -> to_out + residual, C = 320, 8 heads of 40.

  python tools/xattn_block_bench.py [--rows 16 8] [--reps 24] [--pool 12]

Every repetition works on another of `--pool` (x, weights, K / V) sets so that nothing is in L2 when a chain starts."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import ops  # noqa: E402
from sketch2img_amd.unet import pack_xattn_kv, pack_xattn_weights  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, nargs="+", default=[16, 8])
    ap.add_argument("--hw", type=int, default=4096)
    ap.add_argument("--reps", type=int, default=24)
    ap.add_argument("--pool", type=int, default=12)
    a = ap.parse_args()
    d = torch.device("cuda:0")
    C, heads, dh, Lp, L = 320, 8, 40, 80, 77
    scale = dh ** -0.5
    g = torch.Generator().manual_seed(0)
    for rows in a.rows:
        M = rows * a.hw
        sets = []
        for i in range(a.pool):
            wq = (torch.randn(C, C, generator=g) * C ** -0.5).half()
            wo = (torch.randn(C, C, generator=g) * C ** -0.5).half()
            K = torch.randn(rows * Lp, C, generator=g).half().to(d)
            V = torch.randn(rows * Lp, C, generator=g).half().to(d)
            sets.append(dict(wq=wq.to(d), wo=wo.to(d), K=K, V=V, wp=pack_xattn_weights(wq, wo, heads, d),
                             kvp=pack_xattn_kv(K, V, rows, Lp, L, heads), bo=torch.zeros(C, device=d, dtype=torch.float16),
                             gam=torch.ones(C, device=d, dtype=torch.float16), bet=torch.zeros(C, device=d, dtype=torch.float16),
                             x=torch.randn(M, C, device=d, dtype=torch.float16)))
        out = torch.empty(M, C, device=d, dtype=torch.float16)
        a2, q2, o2 = (torch.empty(M, C, device=d, dtype=torch.float16) for _ in range(3))

        def fused(i):
            s = sets[i % a.pool]
            ops.xattn_block(s["x"], a.hw, heads, L, s["gam"], s["bet"], 1e-5, s["wp"], s["kvp"], s["bo"], scale, out=out)

        def four(i):
            s = sets[i % a.pool]
            ops.layernorm(s["x"], s["gam"], s["bet"], 1e-5, out=a2)
            ops.gemm(a2, s["wq"], q2)
            ops.attn_fwd(q2, s["K"], s["V"], rows, heads, a.hw, L, Lp, dh, scale, out=o2, v_rows=True)
            ops.gemm(o2, s["wo"], out, bias=s["bo"], residual=s["x"])

        res = {}
        for name, fn in (("four launches", four), ("fused", fused), ("four launches", four), ("fused", fused)):
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
        t4, tf = min(res["four launches"]), min(res["fused"])
        print(f"rows {rows:3d} (M {M}): four launches {t4:7.1f} us   fused {tf:7.1f} us   x{t4 / tf:.2f}   "
              f"runs {['%.1f' % v for v in res['four launches']]} / {['%.1f' % v for v in res['fused']]}", flush=True)


if __name__ == "__main__":
    main()

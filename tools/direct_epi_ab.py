#!/usr/bin/env python3
"""A/B of gemm2.hip's register-direct epilogue (SKG_DIRECT_EPI, read once per process) against the LDS-staged one on the
plain GEMM shapes of a config-2 batch: one subprocess per setting, alternating, same box.  Every output of a direct run is
compared BIT FOR BIT with the staged run's (same K loop, same fp32 epilogue arithmetic, one rounding: they must be equal).
    python tools/direct_epi_ab.py [settings ...]      default: 0 1 0 1"""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# (M, N, K, residual, relu): the K <= 640 launches of the 64 x 64 / 32 x 32 levels and a few longer ones
SHAPES = [(65536, 960, 320, False, False), (65536, 320, 320, True, False), (65536, 320, 320, False, False), (32768, 320, 320, False, False),
          (32768, 320, 320, True, False), (65536, 320, 640, False, False), (16384, 640, 640, True, False), (16384, 640, 640, False, False),
          (16384, 1920, 640, False, False), (8192, 640, 640, False, False), (65530, 320, 320, True, True), (16384, 640, 2560, True, False),
          (4096, 1280, 1280, True, False), (65536, 320, 1280, True, False)]


def worker():
    import torch
    sys.path.insert(0, ROOT)
    from sketch2img_amd import ops
    from sketch2img_amd._lib import lib
    dev = "cuda:0"
    tag = os.environ.get("SKG_DIRECT_EPI", "0")
    g = torch.Generator().manual_seed(11)
    for M, N, K, use_res, relu in SHAPES:
        a = torch.randn(M, K, generator=g).half().to(dev)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(dev)
        b = torch.randn(N, generator=g).half().to(dev)
        r = torch.randn(M, N, generator=g).half().to(dev) if use_res else None
        out = torch.empty(M, N, device=dev, dtype=torch.float16)
        fn = lambda: ops.gemm(a, w, out=out, bias=b, residual=r, alpha=0.5, relu=relu)
        fn()
        torch.cuda.synchronize()
        ref = 0.5 * (a[:4096].float() @ w.float().t() + b.float())
        if use_res:
            ref = ref + r[:4096].float()
        if relu:
            ref = torch.relu(ref)
        err = float((out[:4096].float() - ref).norm() / ref.norm())
        h = hashlib.sha256(out.cpu().numpy().tobytes()).hexdigest()[:16]
        best = 1e9
        for _ in range(5):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / 20 * 1e3)
        byt = (M * K + N * K + M * N * (2 if use_res else 1)) * 2
        print(f"ROW|{tag}|M{M} N{N} K{K}{'+res' if use_res else ''}{'+relu' if relu else ''} v{lib.skg_gemm_variant(M, N, K, 0, 0)}|{best:.1f}|"
              f"{2.0 * M * N * K / best / 1e6:.1f} TF/s {byt / best / 1e6:.2f} TB/s|{err:.2e}|{h}", flush=True)


if __name__ == "__main__":
    if os.environ.get("SKG_DEPI_WORKER"):
        worker()
    else:
        rows = {}
        for v in (sys.argv[1:] or ["0", "1", "0", "1"]):
            r = subprocess.run([sys.executable, __file__], env=dict(os.environ, SKG_DIRECT_EPI=v, SKG_DEPI_WORKER="1"), capture_output=True, text=True)
            if r.returncode:
                print(r.stderr[-1500:])
            for ln in r.stdout.splitlines():
                if ln.startswith("ROW|"):
                    f = ln.split("|")
                    rows.setdefault(f[2], {}).setdefault(f[1], []).append((float(f[3]), f[6], f[5]))
        print(f"{'shape':34s} " + "  ".join(f"{'EPI=' + k:>22s}" for k in sorted({k for d in rows.values() for k in d})) + "   outputs")
        for shape, d in rows.items():
            keys = sorted(d)
            cells = [f"{min(t for t, _, _ in d[k]):8.1f} us ({'/'.join(f'{t:.1f}' for t, _, _ in d[k])})" for k in keys]
            shas = {k: {h for _, h, _ in d[k]} for k in keys}
            same = len(set.union(*shas.values())) == 1
            print(f"{shape:34s} " + "  ".join(f"{c:>22s}" for c in cells) + ("   bit-equal" if same else f"   DIFFER {shas}") + f"  rel {d[keys[0]][0][2]}")

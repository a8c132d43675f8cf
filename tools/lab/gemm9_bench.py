#!/usr/bin/env python3
"""Pricing run of the v9 kernel (gemm9.hip: hand-placed K loop) against what ships, on conv-shaped problems: one
subprocess per setting of SKG_GEMM9 (read once per process), every output checked against an fp32 torch reference and
for bit-repeatability (20 launches, all equal to the first).
    python tools/lab/gemm9_bench.py [settings ...]        (default: 0 100 101 102 103 300 301 0 100)
A setting "108" / "116" / "124" is a wrong-result probe of the lab build (no DMA / no fragment reads / neither)."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
# (Cin, Cout, H = W): the three problems VERDICT r4 names + the other 64 x 64 / 32 x 32 convolutions of SD1.5
CONVS = [(320, 320, 64), (640, 640, 32), (1280, 1280, 16), (640, 320, 64), (960, 320, 64), (1280, 640, 32), (1920, 640, 32)]
GEMMS = [(65536, 320, 2880, False), (16384, 640, 5760, False), (65536, 320, 1280, True), (16384, 640, 2560, True),
         (65536, 320, 320, True), (65530, 320, 1280, True)]


def worker():
    import torch
    import torch.nn.functional as F
    sys.path.insert(0, ROOT)
    from sketch2img_amd import ops
    from sketch2img_amd._lib import lib
    dev, rows = "cuda:0", 16
    tag = os.environ.get("SKG_GEMM9", "0")
    probe = int(tag) % 100 >= 8

    def timeit(fn, iters=20):
        best = 1e9
        for _ in range(3):
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(iters):
                fn()
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1) / iters * 1e3)
        return best

    g = torch.Generator(device="cpu").manual_seed(3)
    for cin, cout, hw in CONVS:
        x = torch.randn(rows, cin, hw, hw, generator=g).half()
        w = (torch.randn(cout, cin, 3, 3, generator=g) * (9 * cin) ** -0.5).half()
        b = torch.randn(cout, generator=g).half()
        res = torch.randn(rows * hw * hw, cout, generator=g).half().to(dev)
        xn = x.permute(0, 2, 3, 1).reshape(-1, cin).contiguous().to(dev)
        wp = w.permute(0, 2, 3, 1).reshape(cout, -1).contiguous().to(dev)
        bd = b.to(dev)
        out = ops.conv3x3(xn, wp, rows, hw, hw, 0, bias=bd, residual=res).clone()
        n2 = 2 * hw * hw
        ref = F.conv2d(x[:2].float().to(dev), w.float().to(dev), b.float().to(dev), padding=1)
        ref = ref.permute(0, 2, 3, 1).reshape(-1, cout) + res[:n2].float()
        err = float((out[:n2].float() - ref).norm() / ref.norm())
        emax = float((out[:n2].float() - ref).abs().max())
        ref2 = F.conv2d(x[-1:].float().to(dev), w.float().to(dev), b.float().to(dev), padding=1).permute(0, 2, 3, 1).reshape(-1, cout) + res[-hw * hw:].float()
        err = max(err, float((out[-hw * hw:].float() - ref2).norm() / ref2.norm()))
        emax = max(emax, float((out[-hw * hw:].float() - ref2).abs().max()))
        rep = 0
        for _ in range(20):
            o2 = ops.conv3x3(xn, wp, rows, hw, hw, 0, bias=bd, residual=res)
            rep += int(not torch.equal(o2, out))
        t = timeit(lambda: ops.conv3x3(xn, wp, rows, hw, hw, 0, bias=bd, residual=res))
        fl = 2.0 * rows * hw * hw * cout * 9 * cin
        v = lib.skg_gemm_variant(rows * hw * hw, cout, 9 * cin, cin, 1)
        bad = (err > 1e-3 or emax > 0.05 or rep) and not probe
        print(f"G9={tag} conv {cin:5d}->{cout:4d} @{hw}^2 v{v:5d}: {t:8.1f} us {fl / t / 1e6:7.1f} TF/s rel {err:.2e} max {emax:.1e} unrepeatable {rep}"
              + ("  WRONG" if bad else ""), flush=True)
    for M, N, K, use_res in GEMMS:
        a = torch.randn(M, K, generator=g).half().to(dev)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(dev)
        b = torch.randn(N, generator=g).half().to(dev)
        r = torch.randn(M, N, generator=g).half().to(dev) if use_res else None
        out = ops.gemm(a, w, bias=b, residual=r, alpha=0.5, relu=not use_res).clone()
        err = 0.0
        for sl in (slice(0, 4096), slice(M - 4096, M), slice(M // 2 - 77, M // 2 + 4019)):
            ref = 0.5 * (a[sl].float() @ w.float().t() + b.float())
            ref = ref + r[sl].float() if use_res else torch.relu(ref)
            err = max(err, float((out[sl].float() - ref).norm() / ref.norm()))
        rep = 0
        for _ in range(10):
            rep += int(not torch.equal(ops.gemm(a, w, bias=b, residual=r, alpha=0.5, relu=not use_res), out))
        t = timeit(lambda: ops.gemm(a, w, bias=b, residual=r, alpha=0.5, relu=not use_res))
        v = lib.skg_gemm_variant(M, N, K, 0, 0)
        bad = (err > 1e-3 or rep) and not probe
        print(f"G9={tag} gemm M{M} N{N} K{K}{'+res' if use_res else '+relu'} v{v:5d}: {t:8.1f} us {2.0 * M * N * K / t / 1e6:7.1f} TF/s rel {err:.2e} unrepeatable {rep}"
              + ("  WRONG" if bad else ""), flush=True)


if __name__ == "__main__":
    if os.environ.get("SKG_G9_WORKER"):
        worker()
    else:
        for v in (sys.argv[1:] or ["0", "100", "101", "102", "103", "300", "301", "0", "100"]):
            print(f"--- SKG_GEMM9={v}")
            r = subprocess.run([sys.executable, __file__], env=dict(os.environ, SKG_GEMM9=v, SKG_G9_WORKER="1"),
                               capture_output=True, text=True)
            print(r.stdout, r.stderr[-1500:] if r.returncode else "", flush=True)

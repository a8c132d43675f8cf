#!/usr/bin/env python3
"""A/B of the v8 (256 x 160, ping-pong) kernel against v2 on the shapes it takes: one subprocess per setting of
SKG_GEMM8 (read once per process), every output checked against an fp32 torch reference.
    python tools/gemm8_bench.py"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CONVS = [(320, 320, 64), (640, 320, 64), (960, 320, 64), (640, 640, 32), (1280, 640, 32), (1920, 640, 32)]
GEMMS = [(65536, 320, 1280, True), (65536, 320, 320, True), (65536, 960, 320, False), (65536, 2560, 320, False),
         (16384, 640, 2560, True), (16384, 1920, 640, False), (16384, 640, 640, True), (65530, 320, 1280, True)]


def worker():
    import torch
    import torch.nn.functional as F
    sys.path.insert(0, ROOT)
    from sketch2img_amd import ops
    from sketch2img_amd._lib import lib
    dev, rows = "cuda:0", 16
    tag = os.environ.get("SKG_GEMM8", "0")

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
        out = ops.conv3x3(xn, wp, rows, hw, hw, 0, bias=bd, residual=res)
        ref = F.conv2d(x[:2].float().to(dev), w.float().to(dev), b.float().to(dev), padding=1)
        ref = ref.permute(0, 2, 3, 1).reshape(-1, cout) + res[:2 * hw * hw].float()
        err = float((out[:2 * hw * hw].float() - ref).norm() / ref.norm())
        last = out[-hw * hw:].float()
        ref2 = F.conv2d(x[-1:].float().to(dev), w.float().to(dev), b.float().to(dev), padding=1).permute(0, 2, 3, 1).reshape(-1, cout) + res[-hw * hw:].float()
        err = max(err, float((last - ref2).norm() / ref2.norm()))
        t = timeit(lambda: ops.conv3x3(xn, wp, rows, hw, hw, 0, bias=bd, residual=res))
        fl = 2.0 * rows * hw * hw * cout * 9 * cin
        v = lib.skg_gemm_variant(rows * hw * hw, cout, 9 * cin, cin, 1)
        print(f"G8={tag} conv {cin:5d}->{cout:4d} @{hw}^2 v{v:5d}: {t:8.1f} us {fl / t / 1e6:7.1f} TF/s rel err {err:.2e}" + ("  WRONG" if err > 1e-3 else ""), flush=True)
    for M, N, K, use_res in ([] if os.environ.get("SKG_G8_EXP", "0") != "0" else GEMMS):
        a = torch.randn(M, K, generator=g).half().to(dev)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).half().to(dev)
        b = torch.randn(N, generator=g).half().to(dev)
        r = torch.randn(M, N, generator=g).half().to(dev) if use_res else None
        out = ops.gemm(a, w, bias=b, residual=r, alpha=0.5, relu=not use_res)
        ref = 0.5 * (a[-4096:].float() @ w.float().t() + b.float())
        ref = ref + r[-4096:].float() if use_res else torch.relu(ref)
        err = float((out[-4096:].float() - ref).norm() / ref.norm())
        ref0 = 0.5 * (a[:4096].float() @ w.float().t() + b.float())
        ref0 = ref0 + r[:4096].float() if use_res else torch.relu(ref0)
        err = max(err, float((out[:4096].float() - ref0).norm() / ref0.norm()))
        t = timeit(lambda: ops.gemm(a, w, bias=b, residual=r, alpha=0.5, relu=not use_res))
        v = lib.skg_gemm_variant(M, N, K, 0, 0)
        print(f"G8={tag} gemm M{M} N{N} K{K}{'+res' if use_res else '+relu'} v{v:5d}: {t:8.1f} us {2.0 * M * N * K / t / 1e6:7.1f} TF/s rel err {err:.2e}"
              + ("  WRONG" if err > 1e-3 else ""), flush=True)


if __name__ == "__main__":
    if os.environ.get("SKG_G8_WORKER"):
        worker()
    else:
        for v in (sys.argv[1:] or ["0", "1", "0", "1"]):
            v, e = (v.split(":") + ["0"])[:2]
            print(f"--- SKG_GEMM8={v} SKG_G8_EXP={e}")
            r = subprocess.run([sys.executable, __file__], env=dict(os.environ, SKG_GEMM8=v, SKG_G8_EXP=e, SKG_G8_WORKER="1"),
                               capture_output=True, text=True)
            print(r.stdout, r.stderr[-1500:] if r.returncode else "", flush=True)

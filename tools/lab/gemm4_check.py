"""Run as a script (subprocess of test_gpu_kernels.py::test_persistent_gemm, with SKG_GEMM4=1): large-M DIRECT GEMMs
that take the opt-in persistent wave-specialised kernel (gemm4.hip: >= 256 tiles of 128 x 160, K >= 320), against
torch fp32.  Without the variable the same shapes run on gemm2.hip."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import ops  # noqa: E402

D = "cuda:0"
fails = []


def rnd(*shape, seed=0, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).half()


def check(name, got, ref, tol):
    r = ((got.float().cpu() - ref).norm() / ref.norm()).item()
    print(f"{name:44s} rel {r:.2e}", flush=True)
    if not r < tol:
        fails.append(name)


for (M, N, K, use_res, relu, alpha) in [(33000, 320, 320, True, False, 1.0), (32768, 960, 320, False, False, 1.0),
                                        (40000, 320, 64, True, True, 0.5), (33000, 320, 128, False, True, 1.0),
                                        (20000, 640, 1280, True, False, 1.0), (16384, 2560, 320, False, False, 1.0)]:
    A, B = rnd(M, K, seed=1), rnd(N, K, seed=2, scale=K ** -0.5)
    bias, res = rnd(N, seed=3), rnd(M, N, seed=4)
    out = ops.gemm(A.to(D), B.to(D), bias=bias.to(D), residual=res.to(D) if use_res else None, alpha=alpha, relu=relu)
    ref = alpha * (A.float() @ B.float().t() + bias.float())
    if use_res:
        ref = ref + res.float()
    if relu:
        ref = torch.relu(ref)
    # one fp16 rounding without a residual, two with (tile staged in fp16 before the residual is added)
    check(f"gemm {M}x{N}x{K} res={use_res} relu={relu}", out, ref, 4.5e-4 if use_res else 3e-4)
    # column-slice views (lda / ldb / ldc / ldr > width) and untouched neighbours
    if K == 320 and N == 320:
        Abig, outbig = torch.zeros(M, K + 64, dtype=torch.float16), torch.zeros(M, N + 16, device=D, dtype=torch.float16)
        Abig[:, 32:32 + K] = A
        ops.gemm(Abig.to(D)[:, 32:32 + K], B.to(D), out=outbig[:, 8:8 + N], bias=bias.to(D))
        check("  views", outbig[:, 8:8 + N], A.float() @ B.float().t() + bias.float(), 3e-4)
        assert outbig[:, :8].abs().max() == 0 and outbig[:, 8 + N:].abs().max() == 0
print("FAILED: " + ", ".join(fails) if fails else "ALL OK")
sys.exit(1 if fails else 0)

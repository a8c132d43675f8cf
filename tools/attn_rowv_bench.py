"""Self-attention forward: transpose kernel + V^T path vs. row-major V through the LDS transpose read."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import ops
DEV = "cuda:0"


def t(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


for B, heads, dh, N in ((16, 8, 40, 4096), (8, 8, 40, 4096), (16, 8, 80, 1024), (16, 8, 160, 256), (8, 5, 64, 9216), (8, 10, 64, 2304)):
    C = heads * dh
    qkv = torch.randn(B * N, 3 * C, device=DEV).half()
    o = torch.empty(B * N, C, device=DEV, dtype=torch.float16)
    vt = ops.transpose(qkv[:, 2 * C:])
    sc = dh ** -0.5
    t_tr = t(lambda: ops.transpose(qkv[:, 2 * C:]))
    t_a = t(lambda: ops.attn_fwd(qkv[:, :C], qkv[:, C:2 * C], vt, B, heads, N, N, N, dh, sc, out=o))
    t_r = t(lambda: ops.attn_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, heads, N, N, N, dh, sc, out=o, v_rows=True))
    print(f"B{B} h{heads} d{dh} N{N}: transpose {t_tr:.1f} + attn(V^T) {t_a:.1f} = {t_tr + t_a:.1f} us | attn(row V) {t_r:.1f} us", flush=True)

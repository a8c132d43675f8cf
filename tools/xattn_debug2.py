import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import ops
from sketch2img_amd.unet import pack_xattn_kv, pack_xattn_weights

d = torch.device("cuda:0")
C, heads, dh, Lp, L, rows, HW = 320, 8, 40, 80, 77, 1, 128
M = rows * HW
scale = dh ** -0.5
g = torch.Generator().manual_seed(1)
x = (torch.randn(M, C, generator=g)).half()
gam = torch.ones(C).half(); bet = torch.zeros(C).half(); bo = torch.zeros(C).half()
eye = torch.eye(C).half()
wq = (torch.randn(C, C, generator=g) * C ** -0.5).half()
wp = pack_xattn_weights(wq, eye, heads, d)
K0 = torch.zeros(rows * Lp, C).half()


def run(name, V, K=K0):
    kvp = pack_xattn_kv(K.to(d), V.to(d), rows, Lp, L, heads)
    y = ops.xattn_block(x.to(d), HW, heads, L, gam.to(d), bet.to(d), 1e-5, wp, kvp, bo.to(d), scale)
    att = (y.float() - x.to(d).float()).cpu()                      # [M, C] = o (Wo = I)
    # expected: uniform softmax -> mean over the 77 keys
    exp = V.float().view(Lp, C)[:L].mean(0)                       # [C]
    err = (att - exp).abs()
    h0 = att[:, :dh]                                              # head 0
    bad = (err > 2e-3).float().mean()
    print(f"{name:28s} max err {float(err.max()):.4f}  frac bad {float(bad):.3f}  head0 row0 o[0:40:4] {[round(float(v), 4) for v in h0[0, ::4]]} exp {[round(float(v), 4) for v in exp[:dh:4]]}", flush=True)
    if float(bad) > 0:
        bd = (err > 2e-3).float().mean(0).view(heads, dh)          # which d are bad
        print("      bad fraction per d (head 0):", [round(float(v), 2) for v in bd[0]])
        br = (err > 2e-3).float().mean(1)                          # which rows
        print("      bad fraction per row (first 32 rows):", [round(float(v), 2) for v in br[:32]])


run("V = 1", torch.ones(rows * Lp, C).half())
for k in (0, 1, 3, 4, 15, 16, 31, 32, 47, 63, 64, 76):
    V = torch.zeros(Lp, C).half(); V[k] = 77.0
    run(f"V[key {k}] = 77", V)
for dd in (0, 5, 16, 31, 33):
    V = torch.zeros(Lp, C).half(); V[:, dd::dh] = 1.0
    run(f"V[:, d {dd}] = 1", V)

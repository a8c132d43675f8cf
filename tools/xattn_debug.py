"""Component isolation for skg_xattn_block_f16 (debug): special K / V / weights that switch parts of the chain off."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import ops
from sketch2img_amd.unet import pack_xattn_kv, pack_xattn_weights

d = torch.device("cuda:0")
C, heads, dh, Lp, L, rows, HW = 320, 8, 40, 80, 77, 2, 1024
M = rows * HW
scale = dh ** -0.5
g = torch.Generator().manual_seed(1)
rn = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).half()
x = rn(M, C); gam = torch.ones(C).half(); bet = torch.zeros(C).half(); bo = torch.zeros(C).half()
wq, wo = rn(C, C, sc=C ** -0.5), rn(C, C, sc=C ** -0.5)
K, V = rn(rows * Lp, C), rn(rows * Lp, C)
eye = torch.eye(C).half()


def run(name, wq, wo, K, V, nh=None):
    wp = pack_xattn_weights(wq, wo, heads, d)
    kvp = pack_xattn_kv(K.to(d), V.to(d), rows, Lp, L, heads)
    y = ops.xattn_block(x.to(d), HW, heads, L, gam.to(d), bet.to(d), 1e-5, wp, kvp, bo.to(d), scale)
    a2 = ops.layernorm(x.to(d), gam.to(d), bet.to(d), 1e-5)
    q2 = ops.gemm(a2, wq.to(d))
    o2 = ops.attn_fwd(q2, K.to(d), V.to(d), rows, heads, HW, L, Lp, dh, scale, v_rows=True)
    y4 = ops.gemm(o2, wo.to(d), bias=bo.to(d), residual=x.to(d))
    att = (y.float() - x.to(d).float()), (y4.float() - x.to(d).float())
    rel = float((att[0] - att[1]).norm() / att[1].norm().clamp_min(1e-9))
    # per-column-block error of the attention contribution (which output channels / heads are off)
    e = (att[0] - att[1]).reshape(M, heads, dh).norm(dim=(0, 2)) / att[1].reshape(M, heads, dh).norm(dim=(0, 2)).clamp_min(1e-9)
    er = (att[0] - att[1]).reshape(rows, HW, C).norm(dim=(1, 2)) / att[1].reshape(rows, HW, C).norm(dim=(1, 2)).clamp_min(1e-9)
    print(f"{name:44s} rel(attn part) {rel:.3e}  per head-block {[round(float(v), 3) for v in e]}  per image {[round(float(v), 3) for v in er]}", flush=True)


run("general", wq, wo, K, V)
run("Wo = I (out-proj off)", wq, eye, K, V)
run("K = 0 (uniform softmax: mean of V)", wq, eye, torch.zeros_like(K), V)
run("V = 1 (o = 1), Wo = I", wq, eye, K, torch.ones_like(V))
run("Wq = 0 (q = 0: uniform), general Wo", torch.zeros_like(wq), wo, K, V)
Vk = torch.zeros_like(V); Vk.view(rows, Lp, C)[:, 64:77] = 1.0
run("K = 0, V = 1 on keys 64..76 only, Wo = I", wq, eye, torch.zeros_like(K), Vk)
Vd = torch.zeros_like(V); Vd.view(rows, Lp, heads, dh)[..., 32:] = 1.0
run("K = 0, V = 1 on d 32..39 only, Wo = I", wq, eye, torch.zeros_like(K), Vd)
Vd = torch.zeros_like(V); Vd.view(rows, Lp, heads, dh)[..., 32:] = 1.0
run("K = 0, V = 1 on d 32..39 only, general Wo", wq, wo, torch.zeros_like(K), Vd)

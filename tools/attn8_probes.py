#!/usr/bin/env python3
"""Probe builds of attn_fwd8_kernel (make -C sketch2img_amd/csrc phases; results are wrong, timing only): what does each part cost?
   python tools/attn8_probes.py        # one subprocess per SKG_ATTN8_PROBE value, d = 40, N = 4096, 16 rows x 8 heads"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = """
import os, sys, torch
sys.path.insert(0, %r)
from sketch2img_amd import ops
B, heads, N, dh = 16, 8, 4096, 40
C = heads * dh
qkv = torch.randn(B * N, 3 * C, device="cuda").half()
f = lambda: ops.attn_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, heads, N, N, N, dh, dh ** -0.5, v_rows=True)
for _ in range(3): f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): f()
e1.record(); torch.cuda.synchronize()
print("%%8.1f us" %% (e0.elapsed_time(e1) * 100))
""" % ROOT
NAMES = {0: "as shipped (with phase stamps)", 1: "no exponentials", 2: "no MFMAs", 4: "no staging", 8: "no barrier", 12: "no staging, no barrier",
         16: "no LDS fragment reads", 28: "no staging / barrier / fragment reads", 29: "... and no exponentials", 30: "... and no MFMAs (instead)"}
for pr, name in NAMES.items():
    env = dict(os.environ, SKG_LIB=os.path.join(ROOT, "sketch2img_amd", "libskg_phases.so"), SKG_ATTN8="2", SKG_ATTN8_PROBE=str(pr))
    out = subprocess.run([sys.executable, "-c", CODE], env=env, capture_output=True, text=True)
    print(f"probe {pr:2d}  {name:45s} {out.stdout.strip() or out.stderr.strip()[-200:]}")

#!/usr/bin/env python3
"""Where does a wave of the 8-wave attention forward (attn_fwd8_kernel) spend its cycles?  Needs `make -C sketch2img_amd/csrc phases`
(libskg_phases.so, -DSKG_PHASES: s_memtime sums per wave).  Prints cycles per 64-key tile and wave."""
import ctypes, os, sys
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
os.environ["SKG_LIB"] = os.path.join(ROOT, "sketch2img_amd", "libskg_phases.so")
os.environ["SKG_ATTN8"] = "1"
sys.path.insert(0, ROOT)
from sketch2img_amd import ops  # noqa: E402
L = ctypes.CDLL(os.environ["SKG_LIB"])
L.skg_debug_attn_phases.argtypes = [ctypes.c_void_p, ctypes.c_int]
B, heads, N, dh = (int(x) for x in (sys.argv[1:5] if len(sys.argv) > 4 else (16, 8, 4096, 40)))
C = heads * dh
qkv = torch.randn(B * N, 3 * C, device="cuda").half()
f = lambda: ops.attn_fwd(qkv[:, :C], qkv[:, C:2 * C], qkv[:, 2 * C:], B, heads, N, N, N, dh, dh ** -0.5, v_rows=True)
for _ in range(3):
    f()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); f(); e1.record()
torch.cuda.synchronize()
nw = min(1 << 16, B * heads * (N // 256) * 8)
buf = np.zeros((nw, 8), dtype=np.uint64)
assert L.skg_debug_attn_phases(buf.ctypes.data, nw) == 0
t = buf.astype(np.float64)
t = t[t[:, 6] > 0]
nt = t[:, 6]
print(f"attn fwd8 B{B} h{heads} N{N} d{dh}: {e0.elapsed_time(e1) * 1e3:.1f} us (instrumented), {len(t)} waves sampled, {int(nt[0])} tiles each")
for j, name in enumerate(("block 1: fragment reads, PV MFMAs || lane maxima", "re-base test (+ rare branch)", "block 2: QK^T MFMAs || exp2, pack, relayout",
                          "staging: ds_write + global loads + lgkmcnt", "barrier", "whole iteration")):
    v = t[:, j] / nt
    print(f"   {name:50s} median {np.median(v):8.1f}  p10 {np.percentile(v, 10):8.1f}  p90 {np.percentile(v, 90):8.1f}  cycles per tile")

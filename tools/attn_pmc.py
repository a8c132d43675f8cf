#!/usr/bin/env python3
"""Runs only the d = 40 self-attention forward (16 rows, 8 heads, N = 4096) a few times: the workload for
`rocprofv3 --pmc ... -- python tools/attn_pmc.py` (SQ counters of attn_fwd_kernel<2, 3, 2>)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sketch2img_amd import ops
B, heads, N, dh = 16, 8, 4096, 40
C = heads * dh
qkv = torch.randn(B * N, 3 * C, device="cuda").half()
vt = ops.transpose(qkv[:, 2 * C:])
for _ in range(5):
    ops.attn_fwd(qkv[:, :C], qkv[:, C:2 * C], vt, B, heads, N, N, N, dh, dh ** -0.5)
torch.cuda.synchronize()

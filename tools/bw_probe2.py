#!/usr/bin/env python3
"""Plain HBM write / copy rates of this box (torch fill_ / copy_), the yardstick for the GEMM epilogue's store rate."""
import torch
d = "cuda:0"
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e-3 / n
for mb in (42, 168, 336, 1024):
    x = torch.empty(mb << 20, dtype=torch.uint8, device=d); y = torch.empty_like(x)
    tf = t(lambda: x.fill_(1)); tc = t(lambda: y.copy_(x))
    print(f"{mb:5d} MB: fill {tf*1e6:7.1f} us = {mb*1.048576e6/tf/1e12:5.2f} TB/s write | copy {tc*1e6:7.1f} us = {2*mb*1.048576e6/tc/1e12:5.2f} TB/s r+w")

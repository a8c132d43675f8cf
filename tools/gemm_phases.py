#!/usr/bin/env python3
"""Where does a v2 GEMM workgroup spend its time?  Needs `make -C sketch2img_amd/csrc phases` (libskg_phases.so:
same sources with -DSKG_PHASES, s_memtime stamps per workgroup).  Prints, per shape, the median cycles of
setup / K loop / epilogue slab 0 / slab 1, and how workgroups co-resident on one CU overlap in time.

    python tools/gemm_phases.py [M N K]...
"""
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
L = ctypes.CDLL(os.path.join(ROOT, "sketch2img_amd", "libskg_phases.so"))
L.skg_gemm_f16.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                           ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                           ctypes.c_float, ctypes.c_uint, ctypes.c_void_p]
READ = L.skg_debug_phases
READ.argtypes = [ctypes.c_void_p, ctypes.c_int]


def run(M, N, K, flags=0, res=False):
    d = "cuda:0"
    a = torch.randn(M, K, device=d).half()
    w = (torch.randn(N, K, device=d) * K ** -0.5).half()
    b = torch.randn(N, device=d).half()
    out = torch.empty(M, N, device=d, dtype=torch.float16)
    r = torch.randn(M, N, device=d).half() if res else None
    rp, rld = (r.data_ptr(), N) if res else (None, 0)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(3):
        rc = L.skg_gemm_f16(a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, b.data_ptr(), rp, rld, 1.0,
                            flags, st)
        assert rc == 0, rc
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    L.skg_gemm_f16(a.data_ptr(), K, w.data_ptr(), K, out.data_ptr(), N, M, N, K, b.data_ptr(), rp, rld, 1.0, flags, st)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    nb = min(1 << 15, ((M + 127) // 128) * (N // 160 if N % 160 == 0 else N // 64))
    buf = np.zeros((nb, 16), dtype=np.uint64)
    assert READ(buf.ctypes.data, nb) == 0
    t = buf[:, :5].astype(np.int64)
    ok = t[:, 4] > t[:, 0]
    t = t[ok]
    hw, xcc = buf[ok, 6].astype(np.int64), buf[ok, 7].astype(np.int64)
    cu = ((xcc & 0xf) << 8) | (((hw >> 13) & 7) << 5) | (((hw >> 12) & 1) << 4) | ((hw >> 8) & 0xf)
    d_setup, d_loop, d_s0, d_s1 = t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2], t[:, 4] - t[:, 3]
    span = t[:, 4].max() - t[:, 0].min()
    print("   LDS_ALLOC values:", sorted(set(hex(int(v)) for v in buf[ok, 5]))[:8])
    print(f"gemm {M}x{N}x{K} flags {flags:#x} res {res}: {us:.1f} us, {len(t)} workgroups, {len(np.unique(cu))} CUs, "
          f"span {span} ticks ({span / us:.0f} ticks/us)")
    for name, v in (("setup", d_setup), ("K loop", d_loop), ("slab 0", d_s0), ("slab 1", d_s1),
                    ("total", t[:, 4] - t[:, 0])):
        print(f"   {name:7s} median {int(np.median(v)):7d}  p10 {int(np.percentile(v, 10)):7d}  p90 {int(np.percentile(v, 90)):7d}")
    x = buf[ok].astype(np.int64)
    for name, v in (("exit->all waves out (barrier)", x[:, 8] - x[:, 2]), ("staging write", x[:, 9] - x[:, 8]),
                    ("barrier 2", x[:, 10] - x[:, 9]), ("phase 2 (read, convert, store)", x[:, 11] - x[:, 10]),
                    ("barrier top of slab 1", x[:, 3] - x[:, 11]),
                    ("wave exit skew (max - min)", x[:, 12:16].max(1) - x[:, 12:16].min(1))):
        print(f"      {name:32s} median {int(np.median(v)):6d}  p10 {int(np.percentile(v, 10)):6d}  p90 {int(np.percentile(v, 90)):6d}")
    # overlap on a CU: for each workgroup, fraction of its K loop during which another workgroup on the same CU is
    # also in its K loop
    both, tot = 0, 0
    for c in np.unique(cu)[:64]:
        idx = np.where(cu == c)[0]
        for i in idx:
            for j in idx:
                if i != j:
                    both += max(0, min(t[i, 2], t[j, 2]) - max(t[i, 1], t[j, 1]))
            tot += t[i, 2] - t[i, 1]
    print(f"   K-loop time overlapped by another workgroup's K loop on the same CU: {both / max(tot, 1):.2f}")


if __name__ == "__main__":
    res = "--res" in sys.argv
    args = [int(x) for x in sys.argv[1:] if x != "--res"]
    shapes = [tuple(args[i:i + 3]) for i in range(0, len(args), 3)] or [(65536, 320, 320), (65536, 2560, 320),
                                                                         (65536, 320, 1280), (16384, 640, 2560)]
    for s in shapes:
        run(*s, res=res, flags=int(os.environ.get("PH_FLAGS", "0"), 0))

import os, sys, subprocess
for bn in ("160", "128", "64", "320"):
    env = dict(os.environ, SKG_FORCE_BN=bn)
    out = subprocess.run([sys.executable, "tools/gemm_ablate.py"], env=env, capture_output=True, text=True).stdout
    print("BN", bn)
    for l in out.splitlines():
        if l.startswith("gemm 65536x2560x320") or l.startswith("gemm 65536x320x320") or l.startswith("gemm 16384x5120x640"):
            print("   ", l)

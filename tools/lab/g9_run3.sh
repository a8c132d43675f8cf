#!/bin/bash
# round-5 GPU call 3: (1) MFMA power probe, (2) v9 bench with the corrected epilogue, (3) power / clock samples from rocm-smi
# while gemm8 and gemm9 convolutions loop
export SKG_LIB=$PWD/sketch2img_amd/libskg_lab.so
hipcc --offload-arch=gfx950 -O3 tools/ubench/mfma_power.hip -o /tmp/mfma_power 2>/dev/null && /tmp/mfma_power > gpurun_out/g9r3_mfma_power.txt 2>&1
python tools/lab/gemm9_bench.py 0 100 101 300 0 100 > gpurun_out/g9r3_bench.txt 2>&1
for V in 0 100; do
  ( for i in $(seq 1 12); do rocm-smi --showpower --showclocks 2>/dev/null | grep -E "Power|sclk|mclk|fclk" | tr '\n' ' '; echo; sleep 0.5; done ) > gpurun_out/g9r3_smi_v$V.txt &
  SKG_GEMM9=$V python - <<'PY' > gpurun_out/g9r3_loop_v$V.txt 2>&1
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from sketch2img_amd import ops
g = torch.Generator().manual_seed(1)
rows, hw, cin, cout = 16, 64, 960, 320
x = torch.randn(rows * hw * hw, cin, generator=g).half().cuda()
w = (torch.randn(cout, 9 * cin, generator=g) * (9 * cin) ** -0.5).half().cuda()
torch.cuda.synchronize(); t0 = time.time(); n = 0
while time.time() - t0 < 6.0:
    for _ in range(50): ops.conv3x3(x, w, rows, hw, hw, 0)
    torch.cuda.synchronize(); n += 50
print("launches", n, "us per launch", (time.time() - t0) / n * 1e6)
PY
  wait
done
grep -c WRONG gpurun_out/g9r3_bench.txt

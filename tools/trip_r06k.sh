#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=gpurun_out
timeout 900 python -m pytest tests/test_gpu_configs.py -q -m gpu -s -k "config0_trajectories" > $T/r06k_traj.log 2>&1; echo "trajectory rc=$?"; grep "ACCURACY mode:" $T/r06k_traj.log; tail -3 $T/r06k_traj.log
echo "samples-per-GPU sweep, config 2, one box: python bench.py --samples-per-gpu S --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-box-probe  (value = accuracy mode, fast_fp16_value beside it)" > $T/r06_samples_per_gpu_sweep.txt
for S in 1 2 4 8 12 16 32; do
  timeout 900 python bench.py --samples-per-gpu $S --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-box-probe 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); c = d['config']
        print(f\"S = $S: accuracy mode {d['value']:.3f} images/s ({d['ms_per_step']:.0f} ms / batch, {d['achieved_tflops_per_gpu']:.0f} algorithmic TFLOP/s), all-fp16 {c['fast_fp16_value']:.3f} images/s, finite {d['outputs_finite']}\")" | tee -a $T/r06_samples_per_gpu_sweep.txt
done

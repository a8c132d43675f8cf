#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd" -s > $T/r06e_k.log 2>&1; echo "winograd kernel test rc=$?"; tail -4 $T/r06e_k.log
timeout 600 python tools/wino_bench.py > $T/r06_wino_bench.txt 2> $T/r06e_wb.err; echo "wino bench rc=$?"; cat $T/r06_wino_bench.txt; tail -3 $T/r06e_wb.err
one() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', round(d['value'], 4), 'images/s', round(d['ms_per_step'], 1), 'ms/batch finite', d['outputs_finite'])"; }
for rep in 1 2; do
  for V in 0 1 2; do
    SKG_WINO=$V timeout 300 python bench.py --fast-fp16 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-second-mode --no-box-probe 2>/dev/null | one "fast WINO=$V"
  done
done
for V in 0 1 2; do
  SKG_WINO=$V timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-second-mode --no-box-probe 2>/dev/null | one "tolerance WINO=$V"
done
timeout 1500 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu > $T/r06e_p.log 2>&1; echo "pipeline tests rc=$?"; tail -5 $T/r06e_p.log

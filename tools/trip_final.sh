#!/bin/bash
# Last GPU trip of round 3 (budget-bound): guided trajectory parity with the stashing fused FF launch, the final bench lines,
# the rocprofv3 evidence of config 2.   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/trip_final.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=gpurun_out/final; mkdir -p $T
timeout 300 python -m pytest tests/test_gpu_configs.py -q -x -k "config0" -s > $T/config0.log 2>&1; echo "config0 rc=$?"; grep -E "parity|passed|failed" $T/config0.log | tail -8
timeout 420 python bench.py > $T/bench_c2.json 2> $T/bench_c2.err; echo "bench c2 rc=$? $(grep -o '"value": [0-9.]*' $T/bench_c2.json | head -1)"
timeout 420 bash tools/collect_profiles.sh r03 2 > $T/collect.log 2>&1; echo "collect rc=$?"; tail -12 $T/collect.log
timeout 200 python bench.py --config 4 --no-cpu-baseline > $T/bench_c4.json 2> $T/bench_c4.err; echo "bench c4 rc=$? $(grep -o '"value": [0-9.]*' $T/bench_c4.json | head -1)"
timeout 200 python bench.py --config 5 --no-cpu-baseline > $T/bench_c5.json 2> $T/bench_c5.err; echo "bench c5 rc=$? $(grep -o '"value": [0-9.]*' $T/bench_c5.json | head -1)"
timeout 120 python bench.py --shape-report $T/shapes_c2.txt --no-cpu-baseline > $T/bench_c2_shapes.json 2>/dev/null; head -5 $T/shapes_c2.txt

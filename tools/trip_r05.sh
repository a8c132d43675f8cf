#!/bin/bash
# Round-5 measurement trip: bench lines of configs 2 / 4 / 5 (config 2 with roofline, cpu_baseline and the at-tolerance region), the
# per-shape table, the rocprofv3 evidence of config 2 (default and accuracy mode) and of configs 4 / 5.
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/trip_r05.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=gpurun_out/r05final; mkdir -p $T
timeout 600 python bench.py --steps 3 > $T/bench_c2.json 2> $T/bench_c2.err; echo "bench c2 rc=$? $(grep -o '"value": [0-9.]*' $T/bench_c2.json | head -1)"
timeout 300 python bench.py --config 4 --steps 3 --no-cpu-baseline > $T/bench_c4.json 2> $T/bench_c4.err; echo "bench c4 rc=$? $(grep -o '"value": [0-9.]*' $T/bench_c4.json | head -1)"
timeout 300 python bench.py --config 5 --steps 3 --no-cpu-baseline > $T/bench_c5.json 2> $T/bench_c5.err; echo "bench c5 rc=$? $(grep -o '"value": [0-9.]*' $T/bench_c5.json | head -1)"
timeout 200 python bench.py --shape-report $T/shapes_c2.txt --no-cpu-baseline --fast-fp16 --no-second-mode > $T/bench_c2_shapes.json 2>/dev/null; head -4 $T/shapes_c2.txt
timeout 500 bash tools/collect_profiles.sh r05 2 > $T/collect_c2.log 2>&1; echo "collect c2 rc=$?"; tail -8 $T/collect_c2.log
timeout 500 bash tools/collect_profiles.sh r05acc 2 --residual-fp32 --no-second-mode > $T/collect_acc.log 2>&1; echo "collect acc rc=$?"; tail -4 $T/collect_acc.log
timeout 500 bash tools/collect_profiles.sh r05 4 > $T/collect_c4.log 2>&1; echo "collect c4 rc=$?"; tail -3 $T/collect_c4.log
timeout 500 bash tools/collect_profiles.sh r05 5 > $T/collect_c5.log 2>&1; echo "collect c5 rc=$?"; tail -3 $T/collect_c5.log

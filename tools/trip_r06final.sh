#!/bin/bash
# Round-6 FINAL measurement trip (the code as committed): rocprofv3 evidence of config 2 in both modes, then the bench lines.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=gpurun_out/r06final2; mkdir -p $T
timeout 500 bash tools/collect_profiles.sh r06 2 > $T/collect_c2.log 2>&1; echo "collect c2 rc=$?"; tail -3 $T/collect_c2.log
mkdir -p profiles; cp gpurun_out/profiles_out/r06_cfg2_* profiles/ 2>/dev/null
timeout 500 bash tools/collect_profiles.sh r06fast 2 --fast-fp16 > $T/collect_fast.log 2>&1; echo "collect fast rc=$?"; tail -2 $T/collect_fast.log
timeout 700 python bench.py --steps 5 --warmup 2 --shape-report $T/shapes_c2.txt > $T/bench_c2.json 2> $T/bench_c2.err; echo "bench c2 rc=$? $(grep -o '"value": [0-9.]*' $T/bench_c2.json | head -1)"
timeout 300 python bench.py --fast-fp16 --steps 3 --shape-report $T/shapes_c2_fast.txt --no-cpu-baseline --no-second-mode > $T/bench_c2_fast.json 2>/dev/null; echo "bench c2 fast rc=$? $(grep -o '"value": [0-9.]*' $T/bench_c2_fast.json | head -1)"
timeout 400 python bench.py --config 4 --steps 3 --no-cpu-baseline > $T/bench_c4.json 2> $T/bench_c4.err; echo "bench c4 rc=$? $(grep -o '"value": [0-9.]*' $T/bench_c4.json | head -1)"
timeout 400 python bench.py --config 5 --steps 3 --no-cpu-baseline > $T/bench_c5.json 2> $T/bench_c5.err; echo "bench c5 rc=$? $(grep -o '"value": [0-9.]*' $T/bench_c5.json | head -1)"

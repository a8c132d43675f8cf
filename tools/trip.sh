#!/bin/bash
# One GPU trip that checks a tree end to end (what the driver runs at round end, plus a bench line):
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash tools/trip.sh [tag]'
# Outputs land in gpurun_out/<tag>/ (merged back by gpurun).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=gpurun_out/${1:-trip}
mkdir -p $T
timeout 2400 python -m pytest tests -q -m gpu -x > $T/gpu_suite.log 2>&1; echo "pytest rc=$?"; tail -3 $T/gpu_suite.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $T/smoke.log 2>&1; tail -2 $T/smoke.log
timeout 900 python bench.py > $T/bench.json 2> $T/bench.err; echo "bench rc=$?"; grep -o '"value": [0-9.]*' $T/bench.json | head -1

#!/bin/bash
# round 6, second GPU trip: the half-batch two-stream probe, then the whole GPU suite
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=gpurun_out
timeout 900 python tools/halfbatch_probe.py --reps 2 > $T/r06_halfbatch_probe.txt 2> $T/r06b_hb.err; echo "halfbatch rc=$?"; cat $T/r06_halfbatch_probe.txt; tail -3 $T/r06b_hb.err
timeout 2400 python -m pytest tests -x -q -m gpu --durations=15 > $T/r06b_suite.log 2>&1; echo "suite rc=$?"; tail -25 $T/r06b_suite.log

#!/bin/bash
# rocprofv3 --kernel-trace --stats of one bench command -> gpurun_out/<tag>_kernel_stats.csv     usage: tools/stats_only.sh <tag> <bench args...>
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
rm -rf gpurun_out/${TAG}_trace
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/${TAG}_trace -- python bench.py --no-cpu-baseline --no-roofline --steps 2 --warmup 1 "$@" > gpurun_out/${TAG}_trace.log 2>&1
cp "$(find gpurun_out/${TAG}_trace -name '*kernel_stats.csv' | head -1)" gpurun_out/${TAG}_kernel_stats.csv
rm -rf gpurun_out/${TAG}_trace

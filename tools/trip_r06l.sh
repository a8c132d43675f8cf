#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=gpurun_out
timeout 2700 python -m pytest tests -q -m gpu --durations=10 > $T/r06l_suite.log 2>&1; echo "suite rc=$?"; tail -18 $T/r06l_suite.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2

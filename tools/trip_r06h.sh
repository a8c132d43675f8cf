#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd" > $T/r06h_k.log 2>&1; echo "winograd kernel tests rc=$?"; tail -4 $T/r06h_k.log
timeout 2400 python tools/eps_real_batch.py > $T/r06_eps_real_batch.txt 2> $T/r06h_rb.err; echo "real batch rc=$?"; tail -14 $T/r06_eps_real_batch.txt; tail -3 $T/r06h_rb.err

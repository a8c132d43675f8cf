#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "winograd" > $T/r06g_k.log 2>&1; echo "winograd kernel tests rc=$?"; tail -4 $T/r06g_k.log
timeout 1500 python tools/eps_batch_effect.py 981 21 > $T/r06_eps_batch_effect.txt 2> $T/r06g_be.err; echo "batch effect rc=$?"; tail -18 $T/r06_eps_batch_effect.txt; tail -3 $T/r06g_be.err

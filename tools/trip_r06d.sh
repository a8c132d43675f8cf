#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=gpurun_out
timeout 600 python tools/wino_probe.py > $T/r06_wino_probe.txt 2> $T/r06d_wp.err; echo "wino probe rc=$?"; cat $T/r06_wino_probe.txt; tail -3 $T/r06d_wp.err
timeout 1800 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_api.py tests/test_gpu_configs.py -x -q -m gpu -k "declined or accuracy or residual or hilo or config4 or config5 or torch_dtype or 16_rows" > $T/r06d_tests.log 2>&1; echo "tests rc=$?"; tail -8 $T/r06d_tests.log; grep "parity\] accuracy mode" $T/r06d_tests.log

#!/bin/bash
# round 6, third GPU trip: accuracy-mode variants (plain levels x norm-pair sites), then the tests that walk the accuracy mode
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=gpurun_out
timeout 1500 python tools/eps_norm_pairs.py 8 > $T/r06_eps_variants.txt 2> $T/r06c_np.err; echo "variants rc=$?"; tail -12 $T/r06_eps_variants.txt; tail -3 $T/r06c_np.err
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_api.py tests/test_gpu_configs.py -x -q -m gpu -k "declined or accuracy or residual or hilo or config4 or config5 or torch_dtype" > $T/r06c_tests.log 2>&1; echo "tests rc=$?"; tail -8 $T/r06c_tests.log

set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t11
timeout 1500 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "other_configs" -s > gpurun_out/t11/traj.log 2>&1; echo "rc=$?"
grep "parity\]\|passed\|failed\|Error" gpurun_out/t11/traj.log | tail -12

set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t4
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 3000 python -m pytest tests -q -m gpu -x --durations=15 > gpurun_out/t4/gpu_suite.log 2>&1; echo "suite rc=$?"
tail -30 gpurun_out/t4/gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/t4/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 gpurun_out/t4/smoke.log

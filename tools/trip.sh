set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t10
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/t10/gpu_suite.log 2>&1; echo "suite rc=$?"
tail -6 gpurun_out/t10/gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/t10/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/t10/smoke.log

cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=gpurun_out/t34
mkdir -p $T
timeout 2400 python -m pytest tests -q -m gpu -x > $T/gpu_suite.log 2>&1; echo "rc=$?"; tail -4 $T/gpu_suite.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $T/smoke.log 2>&1; tail -2 $T/smoke.log

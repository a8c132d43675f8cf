set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t9
timeout 1200 python -m pytest tests/test_gpu_configs.py tests/test_gpu_pipeline.py -q -m gpu -k "accuracy or shared" -s > gpurun_out/t9/acc.log 2>&1; echo "rc=$?"
grep "parity\] guided\|parity\] sd15 eps  HIP acc\|passed\|failed\|Error" gpurun_out/t9/acc.log | tail -12

cd $GRAFT_REPO_ROOT
T=gpurun_out/t28
mkdir -p $T
timeout 600 python tools/lab/gemmws_bench.py 2>&1 | grep -v amdgpu.ids | tee $T/bench.log

set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t17
timeout 900 python tools/smallm_bench.py --splits --rounds 3 --iters 20 --out gpurun_out/t17/splits.txt > gpurun_out/t17/splits.log 2>&1; echo rc=$?
tail -3 gpurun_out/t17/splits.log

set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t16
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 3000 python -m pytest tests -q -m gpu > gpurun_out/t16/gpu_suite.log 2>&1; echo "suite rc=$?"
tail -4 gpurun_out/t16/gpu_suite.log
timeout 900 python bench.py --steps 3 --warmup 1 --shape-report gpurun_out/t16/r03_cfg2_shapes.txt > gpurun_out/t16/r03_bench_c2.json 2> gpurun_out/t16/c2.err; echo "c2 rc=$?"
timeout 600 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/t16/r03_bench_c4.json 2> gpurun_out/t16/c4.err; echo "c4 rc=$?"
timeout 600 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/t16/r03_bench_c5.json 2> gpurun_out/t16/c5.err; echo "c5 rc=$?"
grep -o '"value": [0-9.]*' gpurun_out/t16/r03_bench_*.json
timeout 1500 bash tools/collect_profiles.sh r03 4 > gpurun_out/t16/prof4.log 2>&1; echo "prof4 rc=$?"
timeout 1500 bash tools/collect_profiles.sh r03 5 > gpurun_out/t16/prof5.log 2>&1; echo "prof5 rc=$?"

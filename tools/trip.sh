set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t7
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 3000 python -m pytest tests -q -m gpu -x > gpurun_out/t7/gpu_suite.log 2>&1; echo "suite rc=$?"
tail -5 gpurun_out/t7/gpu_suite.log
timeout 300 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -k shared -s 2>&1 | grep "parity\] sd15 shared"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/t7/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/t7/smoke.log
timeout 900 python bench.py --steps 3 --warmup 1 --shape-report gpurun_out/t7/r03_cfg2_shapes.txt > gpurun_out/t7/r03_bench_c2.json 2> gpurun_out/t7/c2.err; echo "c2 rc=$?"
timeout 600 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/t7/r03_bench_c4.json 2> gpurun_out/t7/c4.err; echo "c4 rc=$?"
timeout 600 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline > gpurun_out/t7/r03_bench_c5.json 2> gpurun_out/t7/c5.err; echo "c5 rc=$?"
timeout 600 python bench.py --samples-per-gpu 1 --scheduler dpm --ddim-steps 25 --steps 5 --warmup 2 --no-cpu-baseline --no-roofline > gpurun_out/t7/r03_bench_s1_dpm25.json 2> gpurun_out/t7/s1.err; echo "s1 rc=$?"
grep -o '"value": [0-9.]*' gpurun_out/t7/r03_bench_*.json
timeout 1500 bash tools/collect_profiles.sh r03 2 > gpurun_out/t7/prof2.log 2>&1; echo "prof2 rc=$?"
timeout 1500 bash tools/collect_profiles.sh r03 4 > gpurun_out/t7/prof4.log 2>&1; echo "prof4 rc=$?"
timeout 1500 bash tools/collect_profiles.sh r03 5 > gpurun_out/t7/prof5.log 2>&1; echo "prof5 rc=$?"

set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t2
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "hilo or gemm8 or gemm_plain or forced" -s > gpurun_out/t2/kernels.log 2>&1; echo "kernels rc=$?"
grep "hilo\|passed\|failed\|Error" gpurun_out/t2/kernels.log | tail -12
timeout 1200 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "accuracy_mode or decomposition" -s > gpurun_out/t2/acc.log 2>&1; echo "acc rc=$?"
grep "parity\|passed\|failed\|Error" gpurun_out/t2/acc.log | tail -30
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-guidance > gpurun_out/t2/bench_ng.json 2> gpurun_out/t2/bench_ng.err; echo "ng rc=$?"
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-guidance --residual-fp32 > gpurun_out/t2/bench_ng_hp.json 2> gpurun_out/t2/bench_ng_hp.err; echo "hp rc=$?"
grep -o '"value": [0-9.]*' gpurun_out/t2/bench_ng*.json
tail -3 gpurun_out/t2/bench_ng_hp.err
for S in 1 2 4 8 16 32; do timeout 900 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --samples-per-gpu $S > gpurun_out/t2/sweep_s$S.json 2> gpurun_out/t2/sweep_s$S.err; echo "S=$S rc=$? $(grep -o '"value": [0-9.]*' gpurun_out/t2/sweep_s$S.json)"; done

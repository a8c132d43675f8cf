set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t12
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_configs.py -q -m gpu -k "injected or other_configs or config4 or graph_cache" -s > gpurun_out/t12/inj.log 2>&1; echo "rc=$?"
grep "parity\] sketch\|passed\|failed\|Error" gpurun_out/t12/inj.log | tail -8
for i in 1 2; do
SKG_SHARE_CFG=0 timeout 600 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/t12/b4_off$i.json 2> gpurun_out/t12/off$i.err
SKG_SHARE_CFG=1 timeout 600 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/t12/b4_on$i.json 2> gpurun_out/t12/on$i.err
done
grep -o '"value": [0-9.]*' gpurun_out/t12/b4_*.json

set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t15
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_configs.py tests/test_gpu_api.py -q -m gpu -k "clip or config5 or injected" > gpurun_out/t15/clip.log 2>&1; echo "rc=$?"; tail -3 gpurun_out/t15/clip.log
for i in 1 2; do
SKG_INJ_BATCH=0 timeout 600 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/t15/off$i.json 2> gpurun_out/t15/off$i.err
SKG_INJ_BATCH=1 timeout 600 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/t15/on$i.json 2> gpurun_out/t15/on$i.err
done
grep -o '"value": [0-9.]*' gpurun_out/t15/*.json

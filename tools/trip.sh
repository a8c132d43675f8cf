set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t6
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 1500 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -s > gpurun_out/t6/pipeline.log 2>&1; echo "pipeline rc=$?"
grep "shared\|passed\|failed\|Error" gpurun_out/t6/pipeline.log | tail
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "graph or config0 or two_ranks" > gpurun_out/t6/configs.log 2>&1; echo "configs rc=$?"; tail -3 gpurun_out/t6/configs.log
for i in 1 2; do
SKG_SHARE_CFG=0 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/t6/bench_off$i.json 2> gpurun_out/t6/off$i.err; echo "off rc=$?"
SKG_SHARE_CFG=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/t6/bench_on$i.json 2> gpurun_out/t6/on$i.err; echo "on rc=$?"
done
SKG_SHARE_CFG=0 timeout 600 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/t6/bench4_off.json 2> gpurun_out/t6/off4.err
SKG_SHARE_CFG=1 timeout 600 python bench.py --config 4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/t6/bench4_on.json 2> gpurun_out/t6/on4.err
SKG_SHARE_CFG=0 timeout 600 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/t6/bench5_off.json 2> gpurun_out/t6/off5.err
SKG_SHARE_CFG=1 timeout 600 python bench.py --config 5 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/t6/bench5_on.json 2> gpurun_out/t6/on5.err
grep -o '"value": [0-9.]*' gpurun_out/t6/bench*.json

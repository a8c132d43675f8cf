set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t3
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "attention or hilo" -s > gpurun_out/t3/kernels.log 2>&1; echo "kernels rc=$?"
grep "short-key\|passed\|failed\|Error" gpurun_out/t3/kernels.log | tail -24
timeout 600 python tools/attn_bench.py > gpurun_out/t3/attn_new.log 2>&1; echo rc=$?
SKG_NO_ATTN_SHORT=1 timeout 600 python tools/attn_bench.py > gpurun_out/t3/attn_old.log 2>&1; echo rc=$?
grep "kv   77" gpurun_out/t3/attn_new.log gpurun_out/t3/attn_old.log
timeout 1500 python -m pytest tests/test_gpu_pipeline.py -x -q -m gpu -k "unet or sampler or sd15" > gpurun_out/t3/pipeline.log 2>&1; echo "pipeline rc=$?"
tail -4 gpurun_out/t3/pipeline.log
SKG_NO_ATTN_SHORT=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/t3/bench_old.json 2> gpurun_out/t3/bench_old.err; echo "old rc=$?"
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/t3/bench_new.json 2> gpurun_out/t3/bench_new.err; echo "new rc=$?"
SKG_NO_ATTN_SHORT=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/t3/bench_oldb.json 2> gpurun_out/t3/bench_oldb.err; echo "old rc=$?"
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/t3/bench_newb.json 2> gpurun_out/t3/bench_newb.err; echo "new rc=$?"
grep -o '"value": [0-9.]*' gpurun_out/t3/bench_*.json

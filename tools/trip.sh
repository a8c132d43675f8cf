set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/t1
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "gemm or conv or forced or three" > gpurun_out/t1/kernels.log 2>&1; echo "kernels rc=$?"
tail -5 gpurun_out/t1/kernels.log
timeout 600 python tools/smallm_bench.py --rounds 3 --iters 20 --out gpurun_out/t1/smallm.txt > gpurun_out/t1/smallm.log 2>&1; echo "smallm rc=$?"
tail -4 gpurun_out/t1/smallm.log
SKG_LIB=$PWD/sketch2img_amd/libskg_lab.so timeout 600 python tools/smallm_bench.py --rounds 2 --iters 20 --probes --out gpurun_out/t1/smallm_probes.txt > gpurun_out/t1/smallm_probes.log 2>&1; echo "probes rc=$?"
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "rccl or config0 or graph_cache or graph_replay" -s > gpurun_out/t1/configs.log 2>&1; echo "configs rc=$?"
tail -5 gpurun_out/t1/configs.log
SKG_GEMMK=0 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/t1/bench_gk0.json 2> gpurun_out/t1/bench_gk0.err; echo "b0 rc=$?"
SKG_GEMMK=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/t1/bench_gk1.json 2> gpurun_out/t1/bench_gk1.err; echo "b1 rc=$?"
SKG_GEMMK=0 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > gpurun_out/t1/bench_gk0b.json 2> gpurun_out/t1/bench_gk0b.err; echo "b0b rc=$?"
SKG_GEMMK=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --shape-report gpurun_out/t1/shapes_gk1.txt > gpurun_out/t1/bench_gk1b.json 2> gpurun_out/t1/bench_gk1b.err; echo "b1b rc=$?"
grep -o '"value": [0-9.]*' gpurun_out/t1/bench_gk*.json

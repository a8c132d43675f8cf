cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=gpurun_out/t35
mkdir -p $T
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $T/one8.json 2> $T/one8.err
SKG_BENCH_BACKEND=gloo SKG_BENCH_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29655 bench.py --gpus 2 --samples-per-gpu 4 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $T/two4.json 2> $T/two4.err
SKG_BENCH_BACKEND=gloo SKG_BENCH_DEVICE=0 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29656 bench.py --gpus 2 --samples-per-gpu 8 --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $T/two8.json 2> $T/two8.err
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --samples-per-gpu 16 > $T/one16.json 2> $T/one16.err
timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $T/one8b.json 2> $T/one8b.err
grep -o '"value": [0-9.]*' $T/*.json
tail -2 $T/two4.err

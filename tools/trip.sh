cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=gpurun_out/t33
mkdir -p $T
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -k "forked or accuracy" -s > $T/p.log 2>&1; echo "rc=$?"; tail -4 $T/p.log
for i in 1 2; do
SKG_FORK_GUIDANCE=0 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --residual-fp32 > $T/off$i.json 2> $T/off$i.err
SKG_FORK_GUIDANCE=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --residual-fp32 > $T/on$i.json 2> $T/on$i.err
done
grep -o '"value": [0-9.]*' $T/*.json

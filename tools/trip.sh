set -x
cd $GRAFT_REPO_ROOT
T=gpurun_out/t22
mkdir -p $T
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_pipeline.py -q -m gpu -k "forked or shared_cfg" -s > $T/k.log 2>&1; echo "rc=$?"; tail -5 $T/k.log
for i in 1 2; do
SKG_FORK_GUIDANCE=0 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $T/off$i.json 2> $T/off$i.err
SKG_FORK_GUIDANCE=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $T/on$i.json 2> $T/on$i.err
done
grep -o '"value": [0-9.]*' $T/*.json
tail -3 $T/on1.err

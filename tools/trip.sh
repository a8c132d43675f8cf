cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=gpurun_out/t31
mkdir -p $T
for i in 1 2; do
for s in 0 1 2; do
SKG_STAGGER=$s timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $T/s${s}_$i.json 2> $T/s${s}_$i.err
done
done
grep -o '"value": [0-9.]*' $T/*.json

set -x
cd $GRAFT_REPO_ROOT
T=gpurun_out/t19
mkdir -p $T
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "up2" -s > $T/k.log 2>&1; echo "rc=$?"; grep "conv_up2\|passed\|failed" $T/k.log | tail -9
timeout 1500 python -m pytest tests/test_gpu_api.py -q -m gpu -k "vae or decodes" -s > $T/v.log 2>&1; echo "rc=$?"; grep "parity\|passed\|failed" $T/v.log | tail -12
for i in 1 2; do
for c in 2 4 5; do
SKG_UP2_POLY=0 timeout 600 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $T/c${c}_off$i.json 2> $T/c${c}_off$i.err
SKG_UP2_POLY=1 timeout 600 python bench.py --config $c --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $T/c${c}_on$i.json 2> $T/c${c}_on$i.err
done
done
grep -o '"value": [0-9.]*' $T/*.json
grep -o '"vae_decode[^,]*,[^,]*' $T/c2_*.json

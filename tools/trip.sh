set -x
cd $GRAFT_REPO_ROOT
T=gpurun_out/t21
mkdir -p $T
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "up2 or 4x4s2 or conv" -s > $T/k.log 2>&1; echo "rc=$?"; grep "conv_up2\|conv4x4\|passed\|failed\|Error" $T/k.log | tail -16
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_configs.py -q -m gpu -k "backward or sampler or sd15 or guided or unet" > $T/p.log 2>&1; echo "rc=$?"; tail -4 $T/p.log
for i in 1 2; do
SKG_UP2_SMALL=0 SKG_UP2_DGRAD=0 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $T/base$i.json 2> $T/base$i.err
SKG_UP2_SMALL=1 SKG_UP2_DGRAD=0 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $T/small$i.json 2> $T/small$i.err
SKG_UP2_SMALL=1 SKG_UP2_DGRAD=1 timeout 600 python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline > $T/both$i.json 2> $T/both$i.err
done
grep -o '"value": [0-9.]*' $T/*.json

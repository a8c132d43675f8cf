#!/bin/bash
# The SCALE tier in one command: BASELINE.json's metric at 1 / 2 / 4 / 8 GPUs of one node, back to back (weak scaling:
# 8 samples per GPU; configs[1] at N = 1, configs[2] at N = 8).  One process per GPU, torch.distributed backend nccl
# (= RCCL over xGMI): rank 0's weights are broadcast once, every rank samples and decodes its own images, rank 0 gathers
# the uint8 images (sketch2img_amd/dist.py).  Writes one JSON line per N to $OUT (default gpurun_out/scale).
#   tools/scale.sh [steps] [warmup]
set -u
cd "$(dirname "$0")/.."
STEPS=${1:-5}; WARMUP=${2:-2}; OUT=${OUT:-gpurun_out/scale}
export HSA_ENABLE_IPC_MODE_LEGACY=0
mkdir -p "$OUT"
NGPU=$(python -c 'import torch; print(torch.cuda.device_count())')
for N in 1 2 4 8; do
  if [ "$N" -gt "$NGPU" ]; then echo "skip N=$N (only $NGPU GPUs visible)"; continue; fi
  if [ "$N" -eq 1 ]; then
    python bench.py --gpus 1 --steps "$STEPS" --warmup "$WARMUP" > "$OUT/n1.json" 2> "$OUT/n1.err"
  else
    python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $((29700 + N)) \
      bench.py --gpus "$N" --steps "$STEPS" --warmup "$WARMUP" > "$OUT/n$N.json" 2> "$OUT/n$N.err"
  fi
  echo "N=$N rc=$? $(grep -o '"value": [0-9.]*' "$OUT/n$N.json" | head -1)"
done

#!/bin/bash
# Round-5, third trip: the 5 x 64 instantiation of the fused cross-attention launch (tests + config-5 A/B, both modes) and the deep-ring split-K
# launches (bitwise test + config-2 A/B).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=gpurun_out/r05c; mkdir -p $T
timeout 900 python -m pytest tests -m gpu -q -s -k "xattn_block or fused_blocks_on_pairs or split_k_deep or config5 or three_stage or conv3x3 or gemm_" > $T/tests.txt 2>&1
echo "tests rc=$?"; grep -h "passed\|failed\|FAILED" $T/tests.txt | tail -5; grep -h "parity\] xattn_block.*heads\|xattn_block rows.*vs four" $T/tests.txt | head -12
one() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],4), 'images/s', round(d['ms_per_step'],1), 'ms/batch', 'finite', d.get('outputs_finite'))"; }
for i in 1 2; do
  for V in 0 1; do
    SKG_XATTN_D64=$V python bench.py --config 5 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --fast-fp16 --no-second-mode 2>/dev/null | one "config 5 default XATTN_D64=$V"
  done
done | tee $T/ab_xattn_d64.txt
for V in 0 1; do
  SKG_XATTN_D64=$V python bench.py --config 5 --residual-fp32 --no-second-mode --steps 2 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | one "config 5 accuracy XATTN_D64=$V"
done | tee -a $T/ab_xattn_d64.txt
for i in 1 2; do
  for V in 2 3; do
    SKG_SPLIT_NS=$V python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --fast-fp16 --no-second-mode 2>/dev/null | one "config 2 default SPLIT_NS=$V"
  done
done | tee $T/ab_split_ns.txt

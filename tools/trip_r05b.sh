#!/bin/bash
# Round-5, second trip: the tests of the two new launches (delta inside the dQ launch; the stashing cross-attention launch on pairs)
# and their same-box A/Bs (default mode: SKG_ATTN_DQ_DELTA; accuracy mode: SKG_XATTN_KEEP_HP).
#   /usr/local/graft/bin/gpurun --timeout 1500 -- 'bash tools/trip_r05b.sh'
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=gpurun_out/r05b; mkdir -p $T
timeout 700 python -m pytest tests -m gpu -q -s -k "attention_backward or fused_blocks_on_pairs or xattn_block or accuracy_mode or guided_step or forked or shared_cfg or lgp_forward_backward" > $T/tests.txt 2>&1
echo "tests rc=$?"; tail -3 $T/tests.txt; grep -h "xattn_block_hilo_keep\|dq_delta\|passed\|failed" $T/tests.txt | head
one() { python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', round(d['value'],4), 'images/s', round(d['ms_per_step'],1), 'ms/batch', 'finite', d.get('outputs_finite'))"; }
for i in 1 2; do
  for V in 0 1; do
    SKG_ATTN_DQ_DELTA=$V python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --fast-fp16 --no-second-mode 2>/dev/null | one "default ATTN_DQ_DELTA=$V"
  done
done | tee $T/ab_dq_delta.txt
for i in 1 2; do
  for V in 0 1; do
    SKG_XATTN_KEEP_HP=$V python bench.py --residual-fp32 --no-second-mode --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | one "accuracy XATTN_KEEP_HP=$V"
  done
done | tee $T/ab_xattn_keep_hp.txt

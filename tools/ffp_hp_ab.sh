#!/bin/bash
# same-box A/B, accuracy mode: proj_out inside the fused feed-forward launch on pairs (SKG_FF_PROJ_HP=1, default) against the K-doubled GEMM (0)
for i in 1 2; do
  for T in 0 1; do
    SKG_FF_PROJ_HP=$T python bench.py --residual-fp32 --no-second-mode --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('accuracy FF_PROJ_HP=$T', round(d['value'],4), 'images/s', round(d['ms_per_step'],1), 'ms/batch')"
  done
done
python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --fast-fp16 --no-second-mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default', round(d['value'],4), 'images/s', round(d['ms_per_step'],1), 'ms/batch')"

#!/bin/bash
# same-box A/B of the fused conv2 + conv_shortcut launch (SKG_RES_SC=1, default) against the two launches (0): default and accuracy mode
for i in 1 2; do
  for T in 0 1; do
    SKG_RES_SC=$T python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --fast-fp16 --no-second-mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('default  RES_SC=$T', round(d['value'],4), 'images/s', round(d['ms_per_step'],1), 'ms/batch')"
    SKG_RES_SC=$T python bench.py --residual-fp32 --no-second-mode --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('accuracy RES_SC=$T', round(d['value'],4), 'images/s', round(d['ms_per_step'],1), 'ms/batch')"
  done
done

#!/bin/bash
# same-box A/B of the accuracy mode's upsampler form (SKG_HP_UP_TRIPLE=1: K-tripled pair form, 0: default launch + pair output)
for i in 1 2; do
  for T in 1 0; do
    SKG_HP_UP_TRIPLE=$T python bench.py --residual-fp32 --no-second-mode --steps 3 --warmup 1 --no-cpu-baseline --no-roofline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('TRIPLE=$T', round(d['value'],4), 'images/s', round(d['ms_per_step'],1), 'ms/batch')"
  done
done

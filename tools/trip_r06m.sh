#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=gpurun_out
timeout 2400 python tools/eps_real_batch.py > $T/r06_eps_real_batch_hpw.txt 2> $T/r06m_rb.err; echo "real batch rc=$?"; tail -9 $T/r06_eps_real_batch_hpw.txt; tail -3 $T/r06m_rb.err
one() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', round(d['value'], 4), 'images/s', round(d['ms_per_step'], 1), 'ms/batch finite', d['outputs_finite'])"; }
for rep in 1 2; do
  for V in 0 1; do
    SKG_HP_WINO=$V timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-second-mode --no-box-probe 2>/dev/null | one "tolerance HP_WINO=$V"
  done
done

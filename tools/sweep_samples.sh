#!/bin/bash
# images/s against samples per GPU (config 2), and app.py's operating point (one sample, DPM-Solver++ 2M, 25 steps)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
echo "python bench.py --no-cpu-baseline --no-roofline --fast-fp16 --no-second-mode --samples-per-gpu S   (config 2, one MI355X, one box)"
echo "informational: BASELINE configs[1] fixes 8 samples per GPU"
echo "  S  images/s  ms/batch"
for S in 1 2 4 8 16 32; do
  python bench.py --samples-per-gpu $S --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --fast-fp16 --no-second-mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%3d  %8.3f  %8.1f' % ($S, d['value'], d['ms_per_step']))"
done
python bench.py --samples-per-gpu 1 --scheduler dpm --ddim-steps 25 --steps 5 --warmup 1 --no-cpu-baseline --no-roofline --fast-fp16 --no-second-mode 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('one sample, DPM-Solver++ 2M, 25 steps (what app.py runs): %.3f images/s, %.1f ms per image' % (d['value'], d['ms_per_step']))"

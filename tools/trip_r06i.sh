#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=gpurun_out
timeout 2700 python -m pytest tests -q -m gpu --durations=8 > $T/r06i_suite.log 2>&1; echo "suite rc=$?"; tail -16 $T/r06i_suite.log
grep "parity\] accuracy mode" $T/r06i_suite.log
one() { python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('{'):
        d = json.loads(l); print('$1', round(d['value'], 4), 'images/s', round(d['ms_per_step'], 1), 'ms/batch finite', d['outputs_finite'])"; }
for V in 0 1; do
  SKG_WINO_GN=$V timeout 300 python bench.py --fast-fp16 --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-second-mode --no-box-probe 2>/dev/null | one "fast WINO_GN=$V"
  SKG_WINO_GN=$V timeout 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-roofline --no-second-mode --no-box-probe 2>/dev/null | one "tolerance WINO_GN=$V"
done
timeout 900 python bench.py --steps 4 --warmup 1 > $T/r06_bench_i.json 2> $T/r06i_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06_bench_i.json") if l.startswith("{")][-1])
c = d["config"]
print("value", d["value"], "ms", d["ms_per_step"], {k: c[k] for k in ("mode", "eps_max", "eps_rel", "eps_max_unit_var", "fast_fp16_value", "mode_cost", "box_mfma_tflops", "box_sclk_mhz", "box_power_w")})
r = d["roofline"]
print({k: v for k, v in r.items() if k.endswith("_frac") or k in ("kernel", "frac", "achieved")})
PY

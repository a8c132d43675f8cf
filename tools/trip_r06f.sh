#!/bin/bash
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=gpurun_out
timeout 2400 python -m pytest tests -x -q -m gpu --durations=12 > $T/r06f_suite.log 2>&1; echo "suite rc=$?"; tail -22 $T/r06f_suite.log
grep "parity\] accuracy mode" $T/r06f_suite.log
timeout 900 python bench.py --steps 4 --warmup 1 > $T/r06_bench_f.json 2> $T/r06f_bench.err; echo "bench rc=$?"; python - <<'PY'
import json
d = json.loads([l for l in open("gpurun_out/r06_bench_f.json") if l.startswith("{")][-1])
print(json.dumps({k: d[k] for k in ("value", "ms_per_step", "config")}, indent=0)[:2500])
r = d["roofline"]
print({k: v for k, v in r.items() if k not in ("by_operator", "per_kernel")})
PY

#!/bin/bash
# HBM traffic per kernel for bench.py's roofline.traffic: two separate rocprofv3 --pmc passes (FETCH_SIZE and
# WRITE_SIZE do not fit one pass; no trace domains combined with --pmc) over a 10-step run of the bench workload,
# aggregated per kernel name into profiles/<tag>_hbm_counters.json by tools/pmc_to_json.py.
# Run on the GPU box from the repo root:   bash tools/collect_hbm_counters.sh r01
set -e
TAG=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
CMD="python bench.py --ddim-steps 10 --steps 1 --warmup 0 --no-cpu-baseline --no-roofline --no-vae"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf gpurun_out/pmc_$C
  rocprofv3 --pmc $C --output-format csv -d gpurun_out/pmc_$C -- $CMD > gpurun_out/pmc_$C.log 2>&1
done
python tools/pmc_to_json.py gpurun_out/pmc_FETCH_SIZE gpurun_out/pmc_WRITE_SIZE > gpurun_out/${TAG}_hbm_counters.json
python - <<PY
import json
d = json.load(open("gpurun_out/${TAG}_hbm_counters.json"))
for k, v in sorted(d.items(), key=lambda kv: -(2 * kv[1]["FETCH_SIZE"]["sum"] + kv[1]["WRITE_SIZE"]["sum"]))[:12]:
    n = v["FETCH_SIZE"]["launches"]
    print(f"{k[:60]:60s} launches {n:6d}  MB/launch {(2 * v['FETCH_SIZE']['sum'] + v['WRITE_SIZE']['sum']) * 1024 / n / 1e6:9.2f}")
PY

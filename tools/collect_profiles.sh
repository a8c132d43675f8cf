#!/bin/bash
# rocprofv3 evidence for bench.py's roofline, one BASELINE config per call:
#   1. --kernel-trace --stats of the bench command              -> profiles/<tag>_cfg<C>_kernel_stats.{csv,txt}
#   2. --pmc FETCH_SIZE and --pmc WRITE_SIZE (separate passes)    -> profiles/<tag>_cfg<C>_hbm_counters.json
#   3. --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE -> profiles/<tag>_cfg<C>_mfma_counters.{json,txt}
# (counter passes carry --pmc only: no trace domains).  The counter passes run a 6-step schedule of the same workload
# (per-launch figures do not depend on the step count).  Run on the GPU box from the repo root:
#   bash tools/collect_profiles.sh r06 2              (the headline = accuracy mode; round 6)
#   bash tools/collect_profiles.sh r06fast 2 --fast-fp16   (the all-fp16 mode)
set -e
TAG=${1:-r03}
CFG=${2:-2}
EXTRA=${3:-}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out profiles
OUT=gpurun_out/${TAG}_cfg${CFG}
BENCH="python bench.py --config $CFG --no-cpu-baseline --no-roofline --no-second-mode --no-box-probe $EXTRA"
rm -rf ${OUT}_trace
rocprofv3 --kernel-trace --stats --output-format csv -d ${OUT}_trace -- $BENCH --steps 2 --warmup 1 > ${OUT}_trace.log 2>&1
STATS=$(find ${OUT}_trace -name "*kernel_stats.csv" | head -1)
cp "$STATS" profiles/${TAG}_cfg${CFG}_kernel_stats.csv
python - "$STATS" ${OUT}_trace.log profiles/${TAG}_cfg${CFG}_kernel_stats.txt "$BENCH --steps 2 --warmup 1" <<'PY'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
line = [l for l in open(sys.argv[2]) if l.startswith("{")]
tot = sum(float(r["TotalDurationNs"]) for r in rows)
with open(sys.argv[3], "w") as f:
    f.write(f"rocprofv3 --kernel-trace --stats -- {sys.argv[4]}   (1 warm-up + 2 timed batches)\n")
    if line:
        f.write("bench line of the profiled run: " + line[-1])
    f.write(f"total kernel time {tot / 1e6:.1f} ms over 3 batches = {tot / 3e6:.1f} ms per batch\n\n")
    f.write(f"{'kernel':76s} {'calls':>7s} {'total ms':>10s} {'avg us':>9s} {'%':>6s}\n")
    for r in rows[:70]:
        f.write(f"{r['Name'][:76]:76s} {int(r['Calls']):7d} {float(r['TotalDurationNs']) / 1e6:10.2f} "
                f"{float(r['AverageNs']) / 1e3:9.2f} {float(r['Percentage']):6.2f}\n")
PY
PMC="$BENCH --ddim-steps 6 --steps 1 --warmup 0"
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf ${OUT}_pmc_$C
  rocprofv3 --pmc $C --output-format csv -d ${OUT}_pmc_$C -- $PMC > ${OUT}_pmc_$C.log 2>&1
done
python tools/pmc_agg.py ${OUT}_pmc_FETCH_SIZE ${OUT}_pmc_WRITE_SIZE > profiles/${TAG}_cfg${CFG}_hbm_counters.json
rm -rf ${OUT}_pmc_MFMA
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d ${OUT}_pmc_MFMA -- $PMC > ${OUT}_pmc_MFMA.log 2>&1
python tools/pmc_agg.py ${OUT}_pmc_MFMA > profiles/${TAG}_cfg${CFG}_mfma_counters.json
python tools/mfma_report.py profiles/${TAG}_cfg${CFG}_mfma_counters.json profiles/${TAG}_cfg${CFG}_kernel_stats.csv \
  profiles/${TAG}_cfg${CFG}_hbm_counters.json > profiles/${TAG}_cfg${CFG}_mfma_counters.txt
cat profiles/${TAG}_cfg${CFG}_mfma_counters.txt
# gpurun only merges gpurun_out/ back: leave copies there (copy them into profiles/ in the build container)
mkdir -p gpurun_out/profiles_out && cp profiles/${TAG}_cfg${CFG}_* gpurun_out/profiles_out/
# the raw traces are tens of MB per config and gpurun_out/ may carry 64 MiB back: keep the summaries only
rm -rf ${OUT}_trace ${OUT}_pmc_FETCH_SIZE ${OUT}_pmc_WRITE_SIZE ${OUT}_pmc_MFMA

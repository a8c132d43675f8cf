cd "${GRAFT_REPO_ROOT:-/root/repo}"
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=gpurun_out/ffb5; mkdir -p $T
timeout 1500 python -m pytest tests -q -m gpu -x > $T/gpu_suite.log 2>&1; echo "pytest rc=$?"; tail -3 $T/gpu_suite.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke OK')" > $T/smoke.log 2>&1; tail -2 $T/smoke.log
for c in 2 4; do for f in 0 1 0 1; do SKG_FF_BLOCK=$f timeout 400 python bench.py --config $c --no-cpu-baseline --no-roofline > $T/bench_c${c}_ff${f}.json 2>$T/bench.err; echo "c$c ff$f $(grep -o '"value": [0-9.]*' $T/bench_c${c}_ff${f}.json | head -1)"; done; done
for f in 0 1; do SKG_FF_BLOCK=$f timeout 400 python bench.py --config 5 --no-cpu-baseline --no-roofline > $T/bench_c5_ff${f}.json 2>$T/bench.err; echo "c5 ff$f $(grep -o '"value": [0-9.]*' $T/bench_c5_ff${f}.json | head -1)"; done

cd $GRAFT_REPO_ROOT
export HSA_ENABLE_IPC_MODE_LEGACY=0
T=gpurun_out/t30
mkdir -p $T
timeout 900 python bench.py --config 2 --shape-report $T/r03_cfg2_shapes.txt > $T/r03_bench_c2.json 2> $T/bench_c2.err; echo rc=$?
timeout 900 python bench.py --config 4 > $T/r03_bench_c4.json 2> $T/bench_c4.err; echo rc=$?
timeout 900 python bench.py --config 5 > $T/r03_bench_c5.json 2> $T/bench_c5.err; echo rc=$?
grep -o '"value": [0-9.]*' $T/*.json
tail -3 $T/bench_c2.err

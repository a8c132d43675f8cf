cd $GRAFT_REPO_ROOT
T=gpurun_out/t32
mkdir -p $T
timeout 900 python tools/smallm_bench.py --plain --rounds 3 --iters 20 --pool-mb 640 --out $T/cold.txt > $T/cold.log 2>&1; echo rc=$?
timeout 900 python tools/smallm_bench.py --plain --rounds 3 --iters 20 --pool-mb 0 --out $T/warm.txt > $T/warm.log 2>&1; echo rc=$?
paste <(cut -c1-66 $T/cold.txt) <(cut -c49-66 $T/warm.txt) | head -50
tail -2 $T/cold.txt; tail -2 $T/warm.txt

#!/bin/bash
# round 6, first GPU trip: the new / changed tests, the norm-pair table, one full bench line
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
T=gpurun_out
timeout 900 python -m pytest tests/test_gpu_pipeline.py -x -q -k "declined" -s > $T/r06a_t1.log 2>&1; echo "declined rc=$?"; tail -3 $T/r06a_t1.log
timeout 1500 python -m pytest tests/test_gpu_configs.py -x -q -k "16_rows or two_ranks or rccl or contract_line" -s > $T/r06a_t2.log 2>&1; echo "configs rc=$?"; tail -3 $T/r06a_t2.log
timeout 600 python -m pytest tests/test_gpu_api.py -x -q -k "bench_prints" -s > $T/r06a_t3.log 2>&1; echo "api bench rc=$?"; tail -3 $T/r06a_t3.log
timeout 900 python tools/eps_norm_pairs.py 8 > $T/r06_eps_norm_pairs.txt 2> $T/r06a_np.err; echo "norm pairs rc=$?"; tail -10 $T/r06_eps_norm_pairs.txt
timeout 900 python bench.py --steps 3 --warmup 1 > $T/r06_bench_first.json 2> $T/r06a_bench.err; echo "bench rc=$?"; cut -c1-1500 $T/r06_bench_first.json

#!/bin/bash
# one GPU call of the v9 pricing: bench of the schedule variants, then counter passes (matrix-pipe busy, clock, wave states)
# of a few of them on conv 960 -> 320 @ 64 x 64.   usage: tools/lab/g9_run.sh <tag> "<bench settings>" "<pmc settings>"
TAG=$1; BENCH=$2; PMC=$3
export SKG_LIB=$PWD/sketch2img_amd/libskg_lab.so
python tools/lab/gemm9_bench.py $BENCH > gpurun_out/${TAG}_bench.txt 2>&1
for V in $PMC; do
  SKG_GEMM9=$V bash tools/pmc_kernels.sh ${TAG}_v$V conv64 > gpurun_out/${TAG}_pmc_v$V.txt 2>&1
done
grep -c WRONG gpurun_out/${TAG}_bench.txt

#!/bin/bash
# L2-miss traffic of the 3x3 convolution launches, shape by shape (the bench line's roofline.traffic of the dominant conv instantiation reads
# 1.9 x the algorithmic bytes: which shapes, and is it HBM?).  As tools/gemm_fetch.sh: FETCH_SIZE and WRITE_SIZE in separate --pmc passes
# over six launches of one shape; bytes per launch = (2 x FETCH_SIZE + WRITE_SIZE) x 1024; algorithmic = X + W + residual + Y once each.
#   bash tools/conv_fetch.sh > gpurun_out/r05_conv_fetch.txt
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
SHAPES=${*:-"16:64:320:320 8:64:320:320 16:64:640:320 16:32:640:640 16:32:1280:640 16:32:1920:640 8:32:640:640 16:16:1280:1280 16:16:2560:1280 8:16:1280:1280 16:8:1280:1280 16:8:2560:1280 8:8:1280:1280"}
printf "%-24s %-56s %9s %9s %9s %7s %8s\n" "conv rows:hw:cin:cout" kernel "fetch MB" "write MB" "algo MB" ratio "avg us"
for S in $SHAPES; do
  for C in FETCH_SIZE WRITE_SIZE; do
    D=gpurun_out/cf_${C}
    rm -rf $D
    rocprofv3 --pmc $C --output-format csv -d $D -- python tools/pmc_kernel.py conv:$S > $D.log 2>&1 || tail -3 $D.log
  done
  python tools/pmc_agg.py gpurun_out/cf_FETCH_SIZE gpurun_out/cf_WRITE_SIZE | python -c "
import json, sys
d = json.load(sys.stdin)
f = '$S'.split(':'); rows, hw, cin, cout = int(f[0]), int(f[1]), int(f[2]), int(f[3])
M = rows * hw * hw
algo = 2.0 * (M * cin + cout * 9 * cin + (M * cout if 'res' in f[4:] else 0) + M * cout)
for k, v in d.items():
    if ('gemm' not in k and 'splitk' not in k) or 'FETCH_SIZE' not in v:
        continue
    n = v['FETCH_SIZE']['launches']
    if n < 6:
        continue                      # (the workspace set-up launch)
    fe = 2 * v['FETCH_SIZE']['sum'] / n * 1024
    wr = v['WRITE_SIZE']['sum'] / v['WRITE_SIZE']['launches'] * 1024
    ns = v['_ns']['sum'] / v['_ns']['launches'] if '_ns' in v else 0
    print('%-24s %-56s %9.1f %9.1f %9.1f %7.2f %8.1f' % ('$S', k[:56], fe / 1e6, wr / 1e6, algo / 1e6, (fe + wr) / algo, ns / 1e3))
"
done
rm -rf gpurun_out/cf_FETCH_SIZE gpurun_out/cf_WRITE_SIZE

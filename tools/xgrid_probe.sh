cd $GRAFT_REPO_ROOT
for gn in 1 2 4 8; do echo "== SKG_XGRID_GN=$gn"; SKG_XGRID_GN=$gn bash tools/gemm_fetch.sh 16384:5120:640:geglu 16384:1920:640 4096:3840:1280 65536:960:320 16384:640:2560:res 2>&1 | grep -v "^gemm M"; done
echo "== SKG_STREAM_MB=100000 (no non-temporal output stores)"; SKG_STREAM_MB=100000 bash tools/gemm_fetch.sh 65536:320:320 65536:960:320 16384:1920:640 2>&1 | grep -v "^gemm M"

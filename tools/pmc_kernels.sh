#!/bin/bash
# Where the waves of the hot kernels spend their cycles: two rocprofv3 --pmc passes (8 SQ counters each, no trace
# domains) per kernel shape of tools/pmc_kernel.py -> gpurun_out/profiles_out/<tag>_waves_<shape>.json
set -e
TAG=${1:-r02}
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/profiles_out
A="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE"
B="SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS SQ_LDS_DATA_FIFO_FULL SQ_ACTIVE_INST_VMEM"
C="SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU_TRANS_F32 SQ_VALU_MFMA_COEXEC_CYCLES"
for S in ${2:-attn40 attn64 conv gemm_short gemm_ff1}; do
  for P in A B C; do
    D=gpurun_out/${TAG}_waves_${S}_$P
    rm -rf $D
    rocprofv3 --pmc ${!P} --output-format csv -d $D -- python tools/pmc_kernel.py $S > $D.log 2>&1 || tail -3 $D.log
  done
  python tools/pmc_agg.py gpurun_out/${TAG}_waves_${S}_A gpurun_out/${TAG}_waves_${S}_B gpurun_out/${TAG}_waves_${S}_C > gpurun_out/profiles_out/${TAG}_waves_${S}.json
done
python - <<'PY'
import glob, json
for f in sorted(glob.glob("gpurun_out/profiles_out/*_waves_*.json")):
    d = json.load(open(f))
    for k, v in d.items():
        if "SQ_WAVE_CYCLES" not in v or ("gemm" not in k and "attn" not in k and "ff_block" not in k):
            continue
        wc = v["SQ_WAVE_CYCLES"]["sum"] / v["SQ_WAVE_CYCLES"]["launches"]
        print(f, k[:50])
        for c, x in sorted(v.items()):
            if c != "_ns":
                print(f"    {c:32s} {x['sum'] / x['launches']:16.0f}  {x['sum'] / x['launches'] / wc:8.3f} of wave cycles")
PY

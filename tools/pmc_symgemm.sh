#!/bin/bash
# Stall picture of the sketch-product kernel (rk_symm_gemm): SQ counters in separate rocprofv3 passes
# (usage on the GPU box: bash tools/pmc_symgemm.sh <tag>; then python3 tools/pmc_table.py gpurun_out/<tag>_sq symgemm_kernel)
set -o pipefail
TAG=${1:-sg}
OUT=gpurun_out/${TAG}_sq
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_ANY" \
           "SQ_VALU_MFMA_BUSY_CYCLES SQ_VALU_MFMA_COEXEC_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS" \
           "GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/p$i -o p$i -- python3 tools/bench_symgemm.py --r 512 --reps 1 --fold 16 --check-rows 256 > $OUT/p$i.log 2>&1 && echo "pass $i ok" || exit 1
done

#!/bin/bash
# HBM traffic (FETCH_SIZE, WRITE_SIZE: separate passes) and matrix-core / stall counters of the packed symmetric product
# rk_symm_matmat_f16x2 (block kernel + finishing kernel) at k = 1, 5, 10, 16, n = 200 000.
# usage on the GPU box: bash tools/pmc_symm.sh <tag>; then python3 tools/pmc_symm_table.py gpurun_out/<tag>_symm_pmc
set -o pipefail
TAG=${1:-r04}
OUT=gpurun_out/${TAG}_symm_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for k in 1 5 10 16; do
  i=0
  for set in "FETCH_SIZE" "WRITE_SIZE" "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY" "GRBM_GUI_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 200 rocprofv3 --kernel-trace --pmc $set --output-format csv -d $OUT/k${k}_p$i -o p$i -- python3 tools/bench_symm.py --ks $k --modes f16x2 --reps 2 --check-rows 0 > $OUT/k${k}_p$i.log 2>&1 && echo "k=$k pass $i ok" || exit 1
  done
done

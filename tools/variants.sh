#!/bin/bash
# Build tuning variants of librk_hip.so into build_var/<name>/librk_hip.so:
#   tools/variants.sh name1 "-DRK_SYMM_ITEM_ROWS=2048" name2 "-DRK_SYMM_RB=4" ...
#   FILES="rk_symmblk.hip" tools/variants.sh s8 "-DRK_B2_S=8"     # recompile only these sources, link the rest from the
#                                                                  # main build (rlaopt_amd/csrc/build: run make first)
# run one with RK_HIP_LIBRARY=build_var/<name>/librk_hip.so python tools/microbench.py --what nt
set -e
cd "$(dirname "$0")/.."
SRC=rlaopt_amd/csrc
ALL="rk_skinny.hip rk_kernelmat.hip rk_kernelmfma.hip rk_kernelbf16.hip rk_gemm.hip rk_symm.hip rk_symmblk.hip rk_symgemm.hip rk_normal.hip rk_api.cpp rk_comm.cpp"
while [ $# -ge 2 ]; do
  name=$1; flags=$2; shift 2
  out=build_var/$name; mkdir -p $out
  objs=""; tmp=""
  for f in $ALL; do
    if [ -n "$FILES" ] && ! echo " $FILES " | grep -q " $f "; then
      objs="$objs $SRC/build/${f%.*}.o"   # unchanged translation unit: the main build's object
      continue
    fi
    o=$out/${f%.*}.o
    slp=""  # same per-file flags as the Makefile
    case $f in rk_skinny.hip|rk_symm.hip|rk_symmblk.hip|rk_normal.hip) slp="-fno-slp-vectorize";; esac
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function $slp $flags -x hip -c $SRC/$f -o $o &
    objs="$objs $o"; tmp="$tmp $o"
  done
  wait
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $out/librk_hip.so -ldl
  rm -f $tmp
  echo "built $out/librk_hip.so ($flags)"
done

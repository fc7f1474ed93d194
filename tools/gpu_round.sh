#!/bin/bash
# One GPU-box session: parity tests, clean bench, rocprofv3 kernel stats + PMC passes.
# Usage (on the box, from the repo root): bash tools/gpu_round.sh <tag>
set -o pipefail
TAG=${1:-r01}
PART=${2:-all}   # a: tests, bench, rocprof / PMC of the bench; b: the other benches; d: the sketch kernels only; c: round-4 additions (packed multi-RHS
                 # PMC, write-cost probe, float64 lines, self-launched two-rank rehearsal); all: everything (> 30 minutes)
OUT=gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
if [ "$PART" = "d" ]; then  # the sketch kernels only: parity, timings, SQ counters, the assembly by stage, matrix-free PCG
timeout -k 10 600 python -m pytest tests/test_hip_kernels.py tests/test_hip_scale.py -m gpu -q -x -k "sketch or symm_gemm or nystrom or matrix_free" > $OUT/${TAG}_pytest_sketch.log 2>&1; echo "pytest EXIT $?"; tail -2 $OUT/${TAG}_pytest_sketch.log
timeout -k 10 300 python3 tools/bench_symgemm.py --fold 16 > $OUT/${TAG}_bench_symgemm.log 2>&1 && echo "symgemm ok"
timeout -k 10 300 python3 tools/bench_ksketch.py > $OUT/${TAG}_bench_ksketch.log 2>&1 && echo "ksketch ok"
timeout -k 10 300 python3 tools/profile_nystrom_setup.py > $OUT/${TAG}_nystrom_setup.log 2>&1 && echo "nystrom setup ok"
bash tools/pmc_symgemm.sh ${TAG} > $OUT/${TAG}_pmc_symgemm.log 2>&1 && python3 tools/pmc_table.py $OUT/${TAG}_sq symgemm_ > $OUT/${TAG}_symgemm_sq_counters.txt && echo "pmc symgemm ok"
RK_KERNEL_MODE=matrix_free timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-solve > $OUT/${TAG}_bench_matrix_free.log 2>&1 && echo "matrix-free pcg ok"
exit 0
fi
if [ "$PART" != "b" ]; then
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $OUT/${TAG}_smoke.log 2>&1; tail -1 $OUT/${TAG}_smoke.log
timeout -k 10 800 python -m pytest tests -m gpu -q -x > $OUT/${TAG}_pytest_gpu.log 2>&1; echo "pytest EXIT $?"; tail -3 $OUT/${TAG}_pytest_gpu.log
# (the driver's exact command line)
timeout -k 10 400 python3 bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/${TAG}_bench_n1.log 2>$OUT/${TAG}_bench_n1.err && echo "bench ok" && tail -1 $OUT/${TAG}_bench_n1.log | cut -c1-400
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof -o ${TAG} -- python3 bench.py --no-cpu-baseline --no-solve > $OUT/${TAG}_bench_under_rocprof.log 2>&1 && echo "rocprof stats ok"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -o ${TAG} -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-solve > $OUT/${TAG}_pmc_fetch.log 2>&1 && echo "pmc fetch ok"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -o ${TAG} -- python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-solve > $OUT/${TAG}_pmc_write.log 2>&1 && echo "pmc write ok"
# matrix-core counters of the Nystrom assembly GEMMs (own pass: SQ + GRBM only)
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_mfma -o ${TAG} -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-solve > $OUT/${TAG}_pmc_mfma.log 2>&1 && echo "pmc mfma ok"
fi
if [ "$PART" = "a" ]; then exit 0; fi
if [ "$PART" = "c" ] || [ "$PART" = "all" ]; then
# packed symmetric product, 1 .. 16 right-hand sides: block kernel vs round 3's strip kernel vs the float32 kernels
timeout -k 10 400 python3 tools/bench_symm.py --ks 1,2,4,5,8,10,16 > $OUT/${TAG}_bench_symm.log 2>&1 && echo "symm ok"
bash tools/pmc_symm.sh ${TAG} > $OUT/${TAG}_pmc_symm.log 2>&1 && python3 tools/pmc_symm_table.py $OUT/${TAG}_symm_pmc > $OUT/${TAG}_symm_pmc_table.md && echo "pmc symm ok"
timeout -k 10 300 tools/_build/wr_probe 40 > $OUT/${TAG}_wr_probe.log 2>&1 && echo "wr probe ok"
# float64 secondary lines: float64 storage, and float64 iteration over float32 storage
timeout -k 10 500 python3 bench.py --dtype f64 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_f64.log 2>/dev/null && echo "f64 ok"
timeout -k 10 400 python3 bench.py --dtype f64 --storage f32 --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_f64_f32storage.log 2>/dev/null && echo "f64/f32 ok"
# the multi-rank entry as the driver would call it: bench.py starts its own ranks (two gloo ranks on this one GPU)
RK_BENCH_BACKEND=gloo RK_BENCH_ONE_GPU=1 timeout -k 10 500 python3 bench.py --gpus 2 --steps 20 --warmup 5 > $OUT/${TAG}_bench_gloo_world2_selflaunch.log 2>$OUT/${TAG}_bench_gloo_world2_selflaunch.err && echo "self-launched world 2 ok"
# the sketch product of a matrix-free operator: entries evaluated inside the product; its matrix-core busy counter
timeout -k 10 300 python3 tools/bench_ksketch.py > $OUT/${TAG}_bench_ksketch.log 2>&1 && echo "ksketch ok"
timeout -k 10 200 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_ksketch -o ${TAG} -- python3 tools/bench_ksketch.py --r 512 --reps 1 --only-fused --check-rows 0 > $OUT/${TAG}_pmc_ksketch.log 2>&1 && echo "pmc ksketch ok"
timeout -k 10 200 python3 tools/profile_colsample.py > $OUT/${TAG}_profile_colsample.log 2>&1 && echo "colsample profile ok"
timeout -k 10 200 python3 tools/profile_colsample.py --sketch gauss > $OUT/${TAG}_profile_gauss.log 2>&1 && echo "gauss profile ok"
if [ "$PART" = "c" ]; then exit 0; fi
fi
# the two "next" rows of SURVEY 8(f): ASkotch (config 5) and LSQR (config 4)
timeout -k 10 300 python3 tools/bench_sap.py > $OUT/${TAG}_bench_sap.log 2>&1 && echo "sap ok" && tail -1 $OUT/${TAG}_bench_sap.log | cut -c1-300
timeout -k 10 300 python3 tools/bench_lsqr.py > $OUT/${TAG}_bench_lsqr.log 2>&1 && echo "lsqr ok" && tail -1 $OUT/${TAG}_bench_lsqr.log | cut -c1-300
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_sap -o ${TAG}_sap -- python3 tools/bench_sap.py --steps 10 > $OUT/${TAG}_sap_under_rocprof.log 2>&1 && echo "rocprof sap ok"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/${TAG}_prof_lsqr -o ${TAG}_lsqr -- python3 tools/bench_lsqr.py > $OUT/${TAG}_lsqr_under_rocprof.log 2>&1 && echo "rocprof lsqr ok"
# matrix-free kernel product: the three formulations side by side, then the matrix-core busy counter
timeout -k 10 300 python3 tools/bench_kmatmat.py > $OUT/${TAG}_bench_kmatmat.log 2>&1 && echo "kmatmat ok" && grep -c Gpair $OUT/${TAG}_bench_kmatmat.log
timeout -k 10 300 python3 tools/bench_kmatmat.py --kind matern32 --modes bf16x3,direct --cases 100000x100000x64x1,100000x100000x64x4,5000x500000x50x4g,5000x500000x50x10g >> $OUT/${TAG}_bench_kmatmat.log 2>&1 && echo "kmatmat matern ok"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_kmatmat -o ${TAG} -- python3 tools/bench_kmatmat.py --modes bf16x3 --reps 2 --cases 200000x200000x64x1,200000x200000x32x1,5000x500000x50x10g > $OUT/${TAG}_pmc_kmatmat.log 2>&1 && echo "pmc kmatmat ok"
# many right-hand sides: packed symmetric and row-major streaming kernels
timeout -k 10 300 python3 tools/bench_skinny.py > $OUT/${TAG}_bench_skinny.log 2>&1 && timeout -k 10 300 python3 tools/bench_skinny.py --f64 >> $OUT/${TAG}_bench_skinny.log 2>&1 && echo "skinny ok"
timeout -k 10 200 python3 tools/bench_dense_c1.py > $OUT/${TAG}_bench_dense_c1.log 2>&1 && echo "dense c1 ok" && tail -1 $OUT/${TAG}_bench_dense_c1.log | cut -c1-200
# storage-free PCG (matrix-free operator, symmetric evaluation)
RK_KERNEL_MODE=matrix_free timeout -k 10 300 python3 bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-solve > $OUT/${TAG}_bench_matrix_free.log 2>&1 && echo "matrix-free pcg ok"
# BASELINE config 2 (n = 50 000, d = 32, r = 512 float32) as a bench line of its own
timeout -k 10 300 python3 bench.py --n 50000 --d 32 --steps 200 --warmup 20 --no-cpu-baseline > $OUT/${TAG}_bench_c2.log 2>&1 && echo "c2 ok" && tail -1 $OUT/${TAG}_bench_c2.log | cut -c1-200
# the sketch product from the packed strips (rk_symm_gemm): timing, error, SQ counters; the Nystrom assembly by stage
timeout -k 10 300 python3 tools/bench_symgemm.py --fold 16 > $OUT/${TAG}_bench_symgemm.log 2>&1 && echo "symgemm ok"
timeout -k 10 300 python3 tools/profile_nystrom_setup.py > $OUT/${TAG}_nystrom_setup.log 2>&1 && echo "nystrom setup ok"
bash tools/pmc_symgemm.sh ${TAG} > $OUT/${TAG}_pmc_symgemm.log 2>&1 && python3 tools/pmc_table.py $OUT/${TAG}_sq symgemm_ > $OUT/${TAG}_symgemm_sq_counters.txt && echo "pmc symgemm ok"
# ASkotch end to end (300 iterations, callback_freq 100): residual trajectory
timeout -k 10 300 python3 tools/bench_sap.py --solve > $OUT/${TAG}_sap_solve.log 2>&1 && echo "sap solve ok" && tail -1 $OUT/${TAG}_sap_solve.log | cut -c1-300
# rehearsal of the multi-rank path: two gloo ranks sharing this one GPU (n = 100 000)
RK_BENCH_BACKEND=gloo RK_BENCH_ONE_GPU=1 RK_BENCH_N=100000 timeout -k 10 400 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 20 --warmup 5 --no-solve > $OUT/${TAG}_bench_gloo_world2.log 2>&1 && echo "gloo world 2 ok"
timeout -k 10 120 python3 tools/host_probe.py > $OUT/${TAG}_host_probe.log 2>&1 && echo "host probe ok"
ls -R $OUT | grep -c csv

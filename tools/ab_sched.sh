#!/bin/bash
# Same-box A/B of the scheduler kernels (run ON the GPU box from the repo root): tools/ab_sched.sh <tag> [base-lib]
#   smoke at 2^14, the parity / error / golden GPU tests, then kernel times of the shipped library against the base
#   library (FEC_AB_LIB) at 2^20, 2^19, 2^18 for P-256 and Ed25519 variable base, and the scheduler statistics.
set -u
TAG=${1:?tag}
BASE=${2:-tools/ab/libfecgpu_r03.so}
OUT=gpurun_out/ab_sched_$TAG.txt
: > $OUT
timeout -k 10 180 python tools/quick_perf.py 14 1,2 >> $OUT 2>&1 || { echo "smoke failed"; tail -n 20 $OUT; exit 1; }
timeout -k 10 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_errors.py tests/test_gpu_golden.py tests/test_gpu_full_size.py -x -q -m gpu > gpurun_out/ab_sched_tests_$TAG.log 2>&1 || { echo "tests failed"; tail -n 40 gpurun_out/ab_sched_tests_$TAG.log; exit 1; }
tail -n 2 gpurun_out/ab_sched_tests_$TAG.log
for LOGN in 20 19 18 17; do
  echo "## new library, 2^$LOGN" >> $OUT
  timeout -k 10 180 python tools/quick_perf.py $LOGN 1,2 >> $OUT 2>&1 || { echo "perf new failed"; tail -n 20 $OUT; exit 1; }
  echo "## base library $BASE, 2^$LOGN" >> $OUT
  FEC_AB_LIB=$BASE timeout -k 10 180 python tools/quick_perf.py $LOGN 1,2 >> $OUT 2>&1 || { echo "perf base failed"; tail -n 20 $OUT; exit 1; }
done
echo "## fixed base 2^20 new / base" >> $OUT
timeout -k 10 180 python tools/quick_perf.py 20 1 fixed >> $OUT 2>&1 || exit 1
FEC_AB_LIB=$BASE timeout -k 10 180 python tools/quick_perf.py 20 1 fixed >> $OUT 2>&1 || exit 1
echo "## sched_stats (statistics build)" >> $OUT
timeout -k 10 120 tools/microbench/sched_stats 20 >> $OUT 2>&1 || exit 1
echo "## sched_stub (tasks stubbed out: the scheduler alone)" >> $OUT
timeout -k 10 120 tools/microbench/sched_stub 20 >> $OUT 2>&1 || exit 1
grep -E "^##|curve [12] n=2\^(20|19|18|17)|ms \(wave|batches|claim" $OUT

#!/bin/bash
# PMC + kernel-stats passes for one bench.py workload (run ON the GPU box, from the repo root):
#
#   tools/pmc_passes.sh <workload> <outdir-under-gpurun_out> [log2-batch]
#
# One rocprofv3 run per counter group, counters with --kernel-trace only (MI355X_MICROARCH.md,
# "rocprofv3 PMC slots": 8 SQ slots per pass; FETCH_SIZE and WRITE_SIZE cannot share a pass), then
# one --kernel-trace --stats run.  The program after `--` is python3 itself (no env/bash hop).
# tools/pmc_summarize.py turns the CSVs into the per-launch table that is committed under profiles/.
set -u
WL=${1:?workload}
OUT=${2:?outdir}
LOG2=${3:-20}
ROOT=$(pwd)
mkdir -p "$ROOT/$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # name, counters...
  local name=$1; shift
  timeout -k 10 300 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$ROOT/$OUT/$name" -o "$name" -- \
    python3 "$ROOT/bench.py" --workload "$WL" --log2-batch "$LOG2" --steps 3 --warmup 1 --no-cpu-baseline --no-clock-probe \
    > "$ROOT/$OUT/$name.log" 2>&1 || { echo "pass $name failed"; tail -5 "$ROOT/$OUT/$name.log"; return 1; }
  echo "pass $name ok"
}
run sq_a SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_SALU &&
run sq_b SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_INSTS_VMEM SQ_INSTS_SMEM SQ_ACTIVE_INST_SCA SQ_WAVES SQ_LDS_BANK_CONFLICT &&
run valubusy VALUBusy &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run grbm GRBM_GUI_ACTIVE &&
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$ROOT/$OUT/stats" -o stats -- \
  python3 "$ROOT/bench.py" --workload "$WL" --log2-batch "$LOG2" --steps 10 --warmup 2 --no-cpu-baseline --no-clock-probe \
  > "$ROOT/$OUT/stats.log" 2>&1 && echo "stats ok" && tail -1 "$ROOT/$OUT/stats.log"

#!/bin/bash
# The round's records on ONE box (run ON the GPU box from the repo root):  tools/round_records.sh <tag>
#   GPU suite log, bench line of every workload, PMC + kernel-stats passes of the four dominant kernels.
# Summarise afterwards in the container: tools/pmc_refresh_summarize.sh <tag>; copy what is to be judged into profiles/.
set -u
TAG=${1:?tag}
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/gpu_tests_$TAG.log 2>&1 || { tail -n 30 gpurun_out/gpu_tests_$TAG.log; exit 1; }
tail -n 2 gpurun_out/gpu_tests_$TAG.log
for WL in secp256k1-var secp256k1-fixed ed25519-fixed p256-var secp256k1-double ed25519-var; do
  timeout -k 10 300 python bench.py --workload $WL --steps 10 --warmup 2 > gpurun_out/bench_${TAG}_$WL.json 2> gpurun_out/bench_${TAG}_$WL.err || { echo "bench $WL failed"; tail -n 5 gpurun_out/bench_${TAG}_$WL.err; exit 1; }
  python - gpurun_out/bench_${TAG}_$WL.json <<'PY'
import json, sys
t = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
print("%-18s %8.2f M/s  kernel %.2f ms  frac %.3f  cpu %d cores %.1f k/s (1 thread %.2f k/s)" % (
    t["config"]["curve"] + "-" + t["config"]["kind"], t["value"] / 1e6, t["roofline"]["kernel_ms"], t["roofline"]["frac"],
    t["cpu_baseline"]["cores"], t["cpu_baseline"]["value"] / 1e3, t["cpu_baseline"]["single_thread"]["value"] / 1e3))
PY
done
for WL in ed25519-fixed ed25519-var; do   # the Ed25519 workloads again with scalars below l
  timeout -k 10 300 python bench.py --workload $WL --steps 10 --warmup 2 --scalars-below-l > gpurun_out/bench_${TAG}_${WL}_below_l.json 2> /dev/null || { echo "bench $WL --scalars-below-l failed"; exit 1; }
done
# the PMC passes are a call of their own (gpurun limits a call to 20 minutes): bash tools/pmc_refresh.sh $TAG

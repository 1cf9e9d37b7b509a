#!/bin/bash
# Quick look at one workload's L2-side traffic and kernel time (run ON the GPU box from the repo root):
#   tools/pmc_quick.sh <workload> <kernel-name-substring> <tag>
# Two PMC passes only (FETCH_SIZE, WRITE_SIZE -- they cannot share a pass), three timed launches each; prints bytes
# per launch (FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  tools/pmc_passes.sh is the full set.
set -u
WL=${1:?workload}; NEEDLE=${2:?kernel substring}; TAG=${3:?tag}
ROOT=$(pwd); OUT="$ROOT/gpurun_out/pmcq_$TAG"; mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for C in FETCH_SIZE WRITE_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d "$OUT/$C" -o $C -- \
    python3 "$ROOT/bench.py" --workload "$WL" --steps 3 --warmup 1 --no-cpu-baseline --no-clock-probe > "$OUT/$C.log" 2>&1 || { echo "pass $C failed"; tail -n 5 "$OUT/$C.log"; exit 1; }
done
python3 - "$OUT" "$NEEDLE" <<'PY'
import csv, glob, sys
out, needle = sys.argv[1], sys.argv[2]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    per = {}
    for path in glob.glob(out + "/" + c + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if needle in row["Kernel_Name"] and row["Counter_Name"] == c:
                per[row["Dispatch_Id"]] = per.get(row["Dispatch_Id"], 0.0) + float(row["Counter_Value"])
    tot[c] = sum(per.values()) / max(len(per), 1)
f, w = tot["FETCH_SIZE"] * 1024 * 2, tot["WRITE_SIZE"] * 1024
print("%s: fetch %.1f MB (corrected x2), write %.1f MB, total %.1f MB per launch" % (needle, f / 1e6, w / 1e6, (f + w) / 1e6))
PY

#!/bin/bash
# tools/microbench/l2_write under rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE (run ON the GPU box from the repo root):
#   [L2_KIB=<KiB per workgroup>] bash tools/l2_write_pmc.sh      -> bytes per launch and store flavour
ROOT=$(pwd); cd /tmp && export TMPDIR=/tmp
for C in WRITE_SIZE FETCH_SIZE; do
  timeout -k 10 200 rocprofv3 --pmc $C --kernel-trace --output-format csv -d $ROOT/gpurun_out/l2_write_r04/$C -o x -- $ROOT/tools/microbench/l2_write ${L2_KIB:-8} 100 > $ROOT/gpurun_out/l2_write_r04_$C.log 2>&1 || { echo failed $C; tail -5 $ROOT/gpurun_out/l2_write_r04_$C.log; exit 1; }
done
grep -E "region|mode" $ROOT/gpurun_out/l2_write_r04_WRITE_SIZE.log
python3 - $ROOT/gpurun_out/l2_write_r04 <<'PY'
import csv, glob, sys
rows = {}
for c in ("WRITE_SIZE", "FETCH_SIZE"):
    for path in glob.glob(sys.argv[1] + "/" + c + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(path)):
            if "k_rewrite" in row["Kernel_Name"] and row["Counter_Name"] == c:
                k = (row["Kernel_Name"][:40], c); rows[k] = rows.get(k, 0.0) + float(row["Counter_Value"])
for k in sorted(rows):
    scale = 1024 * (2 if k[1] == "FETCH_SIZE" else 1)
    print("%-42s %-10s %10.1f MB per launch%s" % (k[0], k[1], rows[k] * scale / 1e6, " (x2 correction applied)" if k[1] == "FETCH_SIZE" else ""))
PY

#!/bin/bash
# Same-box timing of several builds of the library (run ON the GPU box): tools/ab_libs.sh <tag> "<logn list>" lib1 lib2 ...
# ("shipped" = forge_ec_amd/libfecgpu.so).  Prints the best of three launches per library, curve and size.
set -u
TAG=${1:?tag}; SIZES=${2:?sizes}; shift 2
OUT=gpurun_out/ab_libs_$TAG.txt
: > $OUT
for LOGN in $SIZES; do
  for LIB in "$@"; do
    echo "## $LIB 2^$LOGN" >> $OUT
    if [ "$LIB" = shipped ]; then
      timeout -k 10 180 python tools/quick_perf.py $LOGN ${CURVES:-1,2} ${MODE:-var} >> $OUT 2>&1 || { echo "failed: $LIB"; tail -n 20 $OUT; exit 1; }
    else
      FEC_AB_LIB=$LIB timeout -k 10 180 python tools/quick_perf.py $LOGN ${CURVES:-1,2} ${MODE:-var} >> $OUT 2>&1 || { echo "failed: $LIB"; tail -n 20 $OUT; exit 1; }
    fi
  done
done
python - $OUT <<'PY'
import re, sys
best, lib = {}, None
for line in open(sys.argv[1]):
    m = re.match(r"## (\S+) 2\^(\d+)", line)
    if m:
        lib = m.group(1); continue
    m = re.match(r"curve (\d) n=2\^(\d+) (?:FIXED )?(\S+): ([\d.]+) ms", line)
    if m:
        key = (int(m.group(2)), int(m.group(1)), lib)
        best[key] = min(best.get(key, 1e9), float(m.group(4)))
for key in sorted(best, key=lambda k: (-k[0], k[1])):
    print("2^%d curve %d %-36s %8.3f ms" % (key[0], key[1], key[2], best[key]))
PY

#!/bin/bash
# A variant of libfecgpu.so for same-box A/B runs: the two scheduler translation units rebuilt with extra flags, the
# other objects taken from the in-tree build.   tools/build_variant.sh <name> [-DFEC_P256_QS=896 -DFEC_ED_PS=896 ...]
#   -> tools/ab/libfecgpu_<name>.so   (time it with FEC_AB_LIB=tools/ab/libfecgpu_<name>.so tools/quick_perf.py ...)
set -eu
NAME=${1:?name}; shift
cd "$(dirname "$0")/.."
python -m forge_ec_amd.build > /dev/null
OBJ=forge_ec_amd/csrc/_obj
TMP=$(mktemp -d)
for TU in kernels_p256 kernels_ed; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC "$@" -c forge_ec_amd/csrc/$TU.hip -o $TMP/$TU.o &
done
wait
OTHERS=$(ls $OBJ/*.o | grep -v "kernels_p256.o\|kernels_ed.o")
mkdir -p tools/ab
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/libfecgpu_$NAME.so $OTHERS $TMP/kernels_p256.o $TMP/kernels_ed.o
rm -rf $TMP
echo tools/ab/libfecgpu_$NAME.so

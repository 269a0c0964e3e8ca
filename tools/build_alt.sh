#!/bin/bash
# Variant build of ONE kernel file next to the product library (for interleaved A/B runs: tools/bench_gemm_pair.py,
# tools/bench_gemm_ab.py load every cim_amd/libcim_hip_alt*.so):
#   tools/build_alt.sh TAG FILE.hip -DFLAG=1 ...   ->  cim_amd/libcim_hip_alt_TAG.so  (other objects from csrc/_obj)
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
TAG=$1; SRC=$2; shift 2
python3 -m cim_amd.build > /dev/null
OBJ=/tmp/alt_${TAG}_$$.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function "$@" -c "$ROOT/cim_amd/csrc/$SRC" -o $OBJ
OTHERS=$(ls $ROOT/cim_amd/csrc/_obj/*.o | grep -v "/$SRC.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/cim_amd/libcim_hip_alt_$TAG.so" $OTHERS $OBJ
rm -f $OBJ
echo "$ROOT/cim_amd/libcim_hip_alt_$TAG.so"

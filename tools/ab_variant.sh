#!/bin/bash
# Build an A/B variant of libssq_hip.so from an alternative source of ONE translation unit:
#   tools/ab_variant.sh <unit> <variant.hip> <out.so>      (other units: the objects of the in-tree build)
# A/B points must share a box: run both under one gpurun call with SSQ_HIP_LIB=<out.so>.
set -e
UNIT=$1; SRC=$2; OUT=$3
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/ssqueezepy_amd/csrc/_obj
TMP=$(mktemp -d)
cp "$SRC" "$TMP/$UNIT.hip"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/opt/rocm/include -I"$ROOT/ssqueezepy_amd/csrc" \
    -Wno-unused-result -ffp-contract=off -c "$TMP/$UNIT.hip" -o "$TMP/$UNIT.o"
OBJS=$(ls $OBJ/*.o | grep -v "/$UNIT.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$OUT" $OBJS "$TMP/$UNIT.o" -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
rm -rf "$TMP"

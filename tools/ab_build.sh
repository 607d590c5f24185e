#!/bin/bash
# Build A/B variants of libssq_hip.so from the in-tree source of ONE unit with extra -D flags:
#   tools/ab_build.sh <unit> <name> [flags...]   -> ssqueezepy_amd/libssq_hip_<name>.so
# (other units: the objects of the in-tree build). A/B points must share a box.
set -e
UNIT=$1; NAME=$2; shift 2
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OBJ=$ROOT/ssqueezepy_amd/csrc/_obj
TMP=$(mktemp -d)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I/opt/rocm/include -I"$ROOT/ssqueezepy_amd/csrc" \
    -Wno-unused-result -ffp-contract=off "$@" -c "$ROOT/ssqueezepy_amd/csrc/$UNIT.hip" -o "$TMP/$UNIT.o"
OBJS=$(ls $OBJ/*.o | grep -v "/$UNIT.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/ssqueezepy_amd/libssq_hip_$NAME.so" $OBJS "$TMP/$UNIT.o" \
    -L/opt/rocm/lib -lrocfft -Wl,-rpath,/opt/rocm/lib
rm -rf "$TMP"
echo built libssq_hip_$NAME.so

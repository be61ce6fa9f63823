#!/bin/bash
# Build a variant of the library with extra -D flags for ONE source file (A/B runs: ADER_HIP_LIB=<path>).  Dev tool.
# usage: tools/build_variant.sh <name> <source.hip> <flags...>   ->  ader_amd/variants/libader_hip_<name>.so
set -e
cd "$(dirname "$0")/.."
NAME=$1; SRC=$2; shift 2
mkdir -p ader_amd/variants/_obj
OBJ=ader_amd/variants/_obj/${NAME}_${SRC%.hip}.o
EX=""; [ "$SRC" = "herding.hip" ] && EX="-ffp-contract=off"; [ "$SRC" = "table_update_x3.hip" ] && EX="-fno-slp-vectorize"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt $EX "$@" -c ader_amd/csrc/$SRC -o $OBJ
OBJS=$(ls ader_amd/csrc/_obj/*.o | grep -v "/${SRC%.hip}.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ader_amd/variants/libader_hip_$NAME.so $OBJS $OBJ
echo ader_amd/variants/libader_hip_$NAME.so

#!/bin/bash
# Experiment of round 6 (review item 7): the float32-grade logit FORWARD with fp16 hi / lo operand pieces on pre-scaled operands
# (-DADER_X3_F16: csrc/lbf_common.h) against the product build's bf16 pieces -- accuracy against fp64 and kernel time, same inputs.
# usage (GPU box, repo root): bash tools/ab_f16x3.sh   -> prints both; the variant library is ader_amd/variants/libader_hip_f16x3.so
set -e
cd "$(dirname "$0")/.."
python -m ader_amd.build > /dev/null
mkdir -p ader_amd/variants/_obj
FLAGS="--offload-arch=gfx950 -O3 -fPIC -std=c++17 -Wall -Wno-unused-function -fhip-fp32-correctly-rounded-divide-sqrt -DADER_X3_F16"
for f in logits_x3 logits_bf16; do
    /opt/rocm/bin/hipcc $FLAGS -c ader_amd/csrc/$f.hip -o ader_amd/variants/_obj/f16x3_$f.o 2> /dev/null &
done
wait
OBJS=$(ls ader_amd/csrc/_obj/*.o | grep -v "/x_" | grep -v "/logits_x3.o" | grep -v "/logits_bf16.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ader_amd/variants/libader_hip_f16x3.so $OBJS ader_amd/variants/_obj/f16x3_logits_x3.o ader_amd/variants/_obj/f16x3_logits_bf16.o
for case in init trained; do
    python tools/ab_f16x3.py $case
    ADER_HIP_LIB=$PWD/ader_amd/variants/libader_hip_f16x3.so python tools/ab_f16x3.py $case
done

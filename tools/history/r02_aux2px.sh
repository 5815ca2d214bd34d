#!/bin/bash
mkdir -p gpurun_out/aux
V=chan_vese_amd/csrc/variants
for n in 2048 4096 4608 5120; do
  OPTS=kernel=3 N=$n REPS=3 STEPS=64 python tools/ab_libs.py chan_vese_amd/csrc/libchanvese_hip.so $V/st0/libchanvese_hip.so $V/st0ld0/libchanvese_hip.so 2>&1 | sed "s/^/$n: /" >> gpurun_out/aux/aux2px.log
done
cat gpurun_out/aux/aux2px.log

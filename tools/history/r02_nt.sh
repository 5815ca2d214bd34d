#!/bin/bash
mkdir -p gpurun_out/aux
V=chan_vese_amd/csrc/variants
: > gpurun_out/aux/nt.log
for n in 6144 5120; do
  OPTS=kernel=3 N=$n REPS=3 STEPS=48 python tools/ab_libs.py chan_vese_amd/csrc/libchanvese_hip.so $V/p0_stnt/libchanvese_hip.so $V/p0_ldnt/libchanvese_hip.so $V/p0_bothnt/libchanvese_hip.so 2>&1 | sed "s/^/$n: /" >> gpurun_out/aux/nt.log
done
cat gpurun_out/aux/nt.log

#!/bin/bash
mkdir -p gpurun_out/aux
for r in 1 2; do for v in base auxst auxall; do
  lib=$PWD/chan_vese_amd/csrc/libchanvese_hip.so; [ $v != base ] && lib=$PWD/chan_vese_amd/csrc/variants/$v/libchanvese_hip.so
  echo "== $v" >> gpurun_out/aux/aux1px.log
  CHANVESE_HIP_LIB=$lib N=6144 REPS=3 STEPS=64 python tools/ab_probe.py "kernel=2" >> gpurun_out/aux/aux1px.log 2>&1
  CHANVESE_HIP_LIB=$lib H=1000 W=1000 REPS=3 python tools/ab_probe.py "kernel=2" >> gpurun_out/aux/aux1px.log 2>&1
done; done
cat gpurun_out/aux/aux1px.log

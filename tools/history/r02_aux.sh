#!/bin/bash
# cache-policy bits on the 1-pixel CSV kernel (C=3 at 4096^2) and the 2-step PM kernel (2048^2, 4096^2): base / stores sc1 / stores sc1 + loads sc0
mkdir -p gpurun_out/aux
for round in 1 2; do
for v in base auxst auxall; do
  lib=$PWD/chan_vese_amd/csrc/libchanvese_hip.so; [ $v != base ] && lib=$PWD/chan_vese_amd/csrc/variants/$v/libchanvese_hip.so
  echo "== $v" >> gpurun_out/aux/aux.log
  CHANVESE_HIP_LIB=$lib C=3 REPS=3 python tools/ab_probe.py "chain=1" >> gpurun_out/aux/aux.log 2>&1
  CHANVESE_HIP_LIB=$lib N=2048 python tools/pm_ab.py "pm_kernel=-1" >> gpurun_out/aux/aux.log 2>&1
  CHANVESE_HIP_LIB=$lib N=4096 python tools/pm_ab.py "pm_kernel=-1" >> gpurun_out/aux/aux.log 2>&1
done
done
cat gpurun_out/aux/aux.log

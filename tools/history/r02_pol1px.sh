#!/bin/bash
mkdir -p gpurun_out/aux
: > gpurun_out/aux/pol1px.log
for hw in "512 512" "1000 1000" "1000 1500" "2000 3000"; do set -- $hw
  H=$1 W=$2 REPS=3 python tools/ab_probe.py "kernel=2,wave_pol=0" "kernel=2,wave_pol=1" >> gpurun_out/aux/pol1px.log 2>&1
done
C=3 N=2048 REPS=3 python tools/ab_probe.py "wave_pol=0" "wave_pol=1" >> gpurun_out/aux/pol1px.log 2>&1
C=3 N=4096 REPS=3 python tools/ab_probe.py "wave_pol=0" "wave_pol=1" >> gpurun_out/aux/pol1px.log 2>&1
cat gpurun_out/aux/pol1px.log

#!/bin/bash
mkdir -p gpurun_out/aux
: > gpurun_out/aux/pol2.log
for hw in "6144 6144" "4320 7680" "8192 8192"; do set -- $hw
  H=$1 W=$2 REPS=3 STEPS=48 python tools/ab_probe.py "kernel=3,wave_pol=0" "kernel=3,wave_pol=1" "kernel=2" >> gpurun_out/aux/pol2.log 2>&1
done
cat gpurun_out/aux/pol2.log

#!/bin/bash
mkdir -p gpurun_out/s12
for hw in "4096 4096" "4112 4080" "4096 4064" "4096 4112" "4096 4128" "4096 4160" "4032 4160" "4096 4032"; do set -- $hw
  H=$1 W=$2 REPS=2 python tools/ab_probe.py wave_cskew=500 2>&1 | tee -a gpurun_out/s12/pitch.log
done

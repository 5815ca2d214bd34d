#!/bin/bash
# round 3, session 40: option scan of the 1-channel 2-pixel kernel at 4096^2 in one context (is any non-default setting ahead?)
set -o pipefail
O=gpurun_out/r3s40; mkdir -p $O
N=4096 REPS=3 STEPS=112 timeout -k 10 800 python tools/ab_probe.py "kernel=3" "kernel=3,wave_sync=0" "kernel=3,wave_cskew=400" "kernel=3,wave_cskew=600" "kernel=3,wave_cskew=300" "kernel=3,wave_pol=0" "kernel=3,wave_pol=2" "kernel=3,wave_prio=0" "kernel=3,wave_prio=2" "kernel=3,far_terms=4" "kernel=3,wave_occupancy=4" "kernel=3,wave_occupancy=5" "kernel=3,wave_cls=0" "kernel=3,wave_sync=0,wave_cskew=400" > $O/scan.log 2>&1; cat $O/scan.log

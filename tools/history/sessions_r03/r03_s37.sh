#!/bin/bash
# round 3, session 37: option scan of the 3-channel 2-pixel kernel at 4096^2 in one context (is any non-default setting ahead?)
set -o pipefail
O=gpurun_out/r3s37; mkdir -p $O
C=3 N=4096 REPS=3 STEPS=112 timeout -k 10 800 python tools/ab_probe.py "kernel=3" "kernel=3,wave_pol=1" "kernel=3,wave_pol=2" "kernel=3,wave_cskew=250" "kernel=3,wave_cskew=130" "kernel=3,wave_occupancy=4" "kernel=3,wave_occupancy=5" "kernel=3,wave_sync=0" "kernel=3,wave_prio=0" "kernel=3,wave_prio=2" "kernel=3,wave_cls=0" "kernel=3,far_terms=4" "kernel=3,strip_rows=44" "kernel=3,strip_rows=52" "kernel=3,strip_rows=64" > $O/scan.log 2>&1; cat $O/scan.log

#!/bin/bash
# round 4, session 25: every run-time knob of the 2-pixel kernel once more at 4096^2 x 1 (the kernel changed: near-form march, SDWA addresses)
set -o pipefail
O=gpurun_out/r4s25; mkdir -p $O
REPS=3 timeout -k 10 800 python tools/ab_probe.py "kernel=3" "kernel=3,wave_sync=0" "kernel=3,wave_prio=0" "kernel=3,wave_prio=2" "kernel=3,wave_prio=3" "kernel=3,wave_cls=0" "kernel=3,far_terms=4" \
  "kernel=3,wave_pol=0" "kernel=3,wave_pol=2" "kernel=3,near_switch=0" "kernel=3,wave_cskew=425" "kernel=3,wave_cskew=575" "kernel=3,wave_cskew=650" "kernel=3,wave_sync=0,wave_cskew=425" "kernel=3,wave_xcd=0" "kernel=3" > $O/scan_c1.log 2>&1; cat $O/scan_c1.log

#!/bin/bash
# round 4, session 17: 3 channels at 4096^2, decisive run: class skew 350-500 with priority scheme 2, no barrier, against the default -- five alternations
set -o pipefail
O=gpurun_out/r4s17; mkdir -p $O
C=3 REPS=5 STEPS=304 timeout -k 10 800 python tools/ab_probe.py "wave_sync=0" "wave_sync=0,wave_cskew=501,wave_prio=2" "wave_sync=0,wave_cskew=350" "wave_sync=0,wave_cskew=350,wave_prio=2" "wave_sync=0,wave_cskew=425,wave_prio=2" "wave_sync=0,wave_cskew=425" > $O/ab_c3_decisive.log 2>&1; cat $O/ab_c3_decisive.log

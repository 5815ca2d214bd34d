#!/bin/bash
# round 4, session 19: class skew of the FP32-state flavours (compute-bound: is the staircase the same?) and of 1 channel once more
set -o pipefail
O=gpurun_out/r4s19; mkdir -p $O
REPS=3 timeout -k 10 600 python tools/ab_probe.py "state=32,wave_cskew=500" "state=32,wave_cskew=0" "state=32,wave_cskew=250" "state=32,wave_cskew=375" "state=32,wave_cskew=625" "state=32,wave_cskew=750" "state=32,wave_cskew=500,wave_sync=0" > $O/ab_c1_f32.log 2>&1; cat $O/ab_c1_f32.log
C=3 REPS=3 timeout -k 10 600 python tools/ab_probe.py "state=32,wave_cskew=500" "state=32,wave_cskew=0" "state=32,wave_cskew=250" "state=32,wave_cskew=600" "state=32,wave_cskew=750" "state=32,wave_cskew=500,wave_sync=1" > $O/ab_c3_f32.log 2>&1; cat $O/ab_c3_f32.log
REPS=3 timeout -k 10 600 python tools/ab_probe.py "state=64,wave_cskew=500" "state=64,wave_cskew=400" "state=64,wave_cskew=600" "state=64,wave_cskew=700" > $O/ab_c1_f64.log 2>&1; cat $O/ab_c1_f64.log

#!/bin/bash
# round 3, session 38: 3-channel kernel at 4096^2: class skew x workgroup barrier combinations in one context
set -o pipefail
O=gpurun_out/r3s38; mkdir -p $O
C=3 N=4096 REPS=3 STEPS=112 timeout -k 10 800 python tools/ab_probe.py "kernel=3" "kernel=3,wave_sync=0" "kernel=3,wave_sync=0,wave_cskew=130" "kernel=3,wave_sync=0,wave_cskew=60" "kernel=3,wave_sync=0,wave_cskew=100" "kernel=3,wave_sync=0,wave_cskew=160" "kernel=3,wave_sync=0,wave_cskew=200" "kernel=3,wave_cskew=100" "kernel=3,wave_cskew=160" "kernel=3,wave_sync=0,wave_cskew=130,far_terms=4" "kernel=3,wave_sync=0,wave_cskew=130,wave_pol=2" > $O/scan.log 2>&1; cat $O/scan.log

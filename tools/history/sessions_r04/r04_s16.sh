#!/bin/bash
# round 4, session 16: 3 channels at 4096^2 -- the wave timeline shows a 17 us spread of wave ends (p10 64 .. max 81 us, older dispatch rounds first);
# class skew x barrier x priority scheme once more, in one context
set -o pipefail
O=gpurun_out/r4s16; mkdir -p $O
C=3 REPS=3 timeout -k 10 600 python tools/ab_probe.py "wave_sync=0" "wave_sync=1" "wave_sync=1,wave_cskew=200" "wave_sync=1,wave_cskew=350" "wave_sync=1,wave_cskew=501" "wave_sync=1,wave_cskew=700" \
  "wave_sync=0,wave_cskew=350" "wave_sync=0,wave_cskew=501" "wave_sync=0,wave_cskew=700" "wave_sync=0,wave_prio=0" "wave_sync=0,wave_prio=2" "wave_sync=0,wave_prio=3" "wave_sync=0,wave_prio=4" \
  "wave_sync=0,wave_cskew=501,wave_prio=2" "wave_sync=1,wave_cskew=501,wave_prio=2" "wave_sync=0" > $O/ab_c3.log 2>&1; cat $O/ab_c3.log

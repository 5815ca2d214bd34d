#!/bin/bash
# round 4, session 15: where does a 3-channel 4096^2 launch spend its extra 14 us?  Wave timelines of the 2-pixel kernel, 1 vs 3 channels
set -o pipefail
O=gpurun_out/r4s15; mkdir -p $O
KERNEL=3 ITERS=40 timeout -k 10 200 python tools/wave_timeline.py > $O/timeline_c1.txt 2>&1; head -12 $O/timeline_c1.txt
C=3 KERNEL=3 ITERS=40 timeout -k 10 200 python tools/wave_timeline.py > $O/timeline_c3.txt 2>&1; head -12 $O/timeline_c3.txt
C=3 KERNEL=3 ITERS=40 timeout -k 10 200 python tools/wave_timeline.py wave_sync=1 wave_cskew=500 > $O/timeline_c3_sync_skew.txt 2>&1; head -12 $O/timeline_c3_sync_skew.txt

#!/bin/bash
# round 4, session 20: profiles/r04_C3 again with the 3-channel class skew (0.425) as the default; wave timeline with it
set -o pipefail
O=gpurun_out/r4s20; mkdir -p $O
bash tools/profile_round.sh r04_C3 "--config C3" > gpurun_out/r04_C3.summary.txt 2>&1; tail -6 gpurun_out/r04_C3.summary.txt | cut -c1-400
C=3 KERNEL=3 ITERS=40 timeout -k 10 200 python tools/wave_timeline.py > $O/timeline_c3_skew425.txt 2>&1; head -5 $O/timeline_c3_skew425.txt

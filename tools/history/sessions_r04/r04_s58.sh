#!/bin/bash
# round 4, session 58: C4 evidence, last collection (Perona-Malik: first poll behind the march barrier)
set -o pipefail
O=gpurun_out/r4s58; mkdir -p $O
bash tools/profile_round.sh r04_C4d "--config C4" > $O/profile_C4.log 2>&1; tail -12 $O/profile_C4.log | cut -c1-400
timeout -k 10 200 python tools/resident_timeline.py > $O/resident_timeline_2048.txt 2>&1
timeout -k 10 200 python tools/pm_resident_timeline.py > $O/pm_resident_timeline_2048.txt 2>&1; tail -3 $O/pm_resident_timeline_2048.txt | cut -c1-250

#!/bin/bash
# round 4, session 27: timelines of the resident kernels with and without the priority by quarters
set -o pipefail
O=gpurun_out/r4s27; mkdir -p $O
for p in 0 1; do
  timeout -k 10 200 python tools/resident_timeline.py res_prio=$p > $O/resident_timeline_2048_prio$p.txt 2>&1; tail -14 $O/resident_timeline_2048_prio$p.txt
  timeout -k 10 200 python tools/pm_resident_timeline.py res_prio=$p > $O/pm_resident_timeline_2048_prio$p.txt 2>&1; tail -10 $O/pm_resident_timeline_2048_prio$p.txt
done

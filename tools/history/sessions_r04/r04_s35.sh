#!/bin/bash
# round 4, session 35: when do the master's eight waves see their shares of the arrivals (diagnostic stamps), three collections
set -o pipefail
O=gpurun_out/r4s35; mkdir -p $O
for i in 1 2; do
timeout -k 10 200 python tools/resident_timeline.py res_go_share=5 > $O/resident_timeline_2048_$i.txt 2>&1; tail -5 $O/resident_timeline_2048_$i.txt
done

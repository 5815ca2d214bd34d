#!/bin/bash
# round 4, session 45: does giving the older wave of a SIMD more rows than the younger balance the march? (generic march, diagnostic res_band_skew)
set -o pipefail
O=gpurun_out/r4s45; mkdir -p $O
N=2048 REPS=4 timeout -k 10 400 python tools/ab_probe.py "resident=1,res_straight=1" "resident=1,res_straight=0" "resident=1,res_straight=0,res_band_skew=1" "resident=1,res_straight=0,res_band_skew=2" "resident=1,res_straight=0,res_band_skew=3" > $O/ab_2048.log 2>&1; cat $O/ab_2048.log
for d in 0 1 2; do
timeout -k 10 200 python tools/resident_timeline.py res_straight=0 res_band_skew=$d > $O/resident_timeline_2048_generic_skew$d.txt 2>&1; grep "tile \|all waves done\|iteration period" $O/resident_timeline_2048_generic_skew$d.txt | cut -c1-250
done

#!/bin/bash
# round 4, session 23: does the class skew suit the batch (8 interleaved contexts) and the all-near regime?
set -o pipefail
O=gpurun_out/r4s23; mkdir -p $O
for o in "wave_cskew=500" "wave_cskew=0" "wave_cskew=250" "wave_cskew=500"; do
  timeout -k 10 200 python bench.py --config C5 --no-cpu-baseline --opt $o > $O/c5_$o.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/c5_$o.json')); r=d['roofline']; print('C5 $o', round(r['avg_launch_us'],2), round(r['frac'],4), round(r['frac_wall'],4), round(d['value']))"
done
for o in "wave_cskew=500" "wave_cskew=250" "wave_cskew=0" "wave_cskew=700" "wave_cskew=500"; do
  timeout -k 10 200 python bench.py --config near --no-cpu-baseline --no-phases --opt $o > $O/near_$o.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/near_$o.json')); r=d['roofline']; print('near $o', round(r['avg_launch_us'],2), round(r['frac'],4), round(d['value']))"
done

#!/bin/bash
# round 4, session 21: the driver's 20-step command read 62.5 us in the validation session (57.9 in the others): what the scratch-context warm-up
# iterates on (a settled level set vs the first iterations of a run), six alternations of fresh processes on one box
set -o pipefail
O=gpurun_out/r4s21; mkdir -p $O
for i in 1 2 3 4 5 6; do
  for m in settled early; do
    timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-phases --prewarm-data $m > $O/b_${m}_$i.json 2> /dev/null
    python -c "
import json; d=json.load(open('$O/b_${m}_$i.json')); r=d['roofline']; print('$m', $i, round(r['avg_launch_us'],2), round(r['frac'],4), round(r['frac_wall'],4))" | tee -a $O/summary.txt
  done
done
timeout -k 10 120 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --prewarm-ms 0 > $O/b_noprewarm.json 2> /dev/null; python -c "
import json; d=json.load(open('$O/b_noprewarm.json')); r=d['roofline']; print('no prewarm', round(r['avg_launch_us'],2), round(r['frac'],4), round(r['frac_wall'],4), d.get('phases'))" | tee -a $O/summary.txt

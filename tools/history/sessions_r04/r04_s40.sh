#!/bin/bash
# round 4, session 40: synchronisation lines and border buffers of the resident kernels in fine-grained / uncached device memory (res_mem 1 / 2)
set -o pipefail
O=gpurun_out/r4s40; mkdir -p $O
for i in 1 2; do for m in 0 1 2; do
  timeout -k 10 200 python bench.py --config C4 --no-cpu-baseline --no-phases --opt res_mem=$m > $O/c4_m${m}_$i.json 2>$O/c4_m${m}_$i.err; python -c "
import json; d=json.load(open('$O/c4_m${m}_$i.json')); print('res_mem=$m', 'csv us/iter', round(d['roofline']['avg_launch_us'],2), 'pm us/step', round(d['pm']['us_per_step'],3), d['checked'])" || tail -3 $O/c4_m${m}_$i.err
done; done
for m in 1 2; do
timeout -k 10 200 python tools/resident_timeline.py res_mem=$m > $O/resident_timeline_2048_mem$m.txt 2>&1; tail -9 $O/resident_timeline_2048_mem$m.txt
done

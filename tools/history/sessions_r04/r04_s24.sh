#!/bin/bash
# round 4, session 24: equal strips for a batch, automatically: the batch test, the C5 line, the interleaved / resident-rule tests
set -o pipefail
O=gpurun_out/r4s24; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "batch or automatic_flow or interleaved or bench" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
for i in 1 2; do timeout -k 10 200 python bench.py --config C5 --no-cpu-baseline > $O/c5_$i.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/c5_$i.json')); r=d['roofline']; print('C5', round(r['avg_launch_us'],2), round(r['frac'],4), round(r['frac_wall'],4), round(d['value']), r['kernel'])"; done
timeout -k 10 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/c2.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/c2.json')); r=d['roofline']; print('C2 driver', round(r['avg_launch_us'],2), round(r['frac'],4), round(r['frac_wall'],4), round(d['value']))"

#!/bin/bash
# round 4, session 52: wider / contiguous border transactions in both resident kernels -- whole GPU suite, fuzz, C4 bench, timelines
set -o pipefail
O=gpurun_out/r4s52; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
grep -q "rc=0" $O/pytest.log || exit 1
CASES=120 SEED=11 timeout -k 10 400 python tools/fuzz_resident.py > $O/fuzz.log 2>&1; tail -2 $O/fuzz.log
for i in 1 2 3; do
  timeout -k 10 200 python bench.py --config C4 --no-cpu-baseline --no-phases > $O/c4_$i.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/c4_$i.json')); print('csv us/iter', round(d['roofline']['avg_launch_us'],2), 'pm us/step', round(d['pm']['us_per_step'],3), d['checked'])"
done
timeout -k 10 200 python tools/resident_timeline.py > $O/resident_timeline_2048.txt 2>&1; tail -9 $O/resident_timeline_2048.txt | cut -c1-250
timeout -k 10 200 python tools/pm_resident_timeline.py > $O/pm_resident_timeline_2048.txt 2>&1; tail -9 $O/pm_resident_timeline_2048.txt | cut -c1-250

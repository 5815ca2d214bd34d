#!/bin/bash
# round 4, session 26: priority by quarters of a band in the resident kernels (the two waves of a SIMD finish together): C4 with and without, alternating
set -o pipefail
O=gpurun_out/r4s26; mkdir -p $O
for i in 1 2 3; do for p in 0 1; do
  timeout -k 10 200 python bench.py --config C4 --no-cpu-baseline --no-phases --opt res_prio=$p > $O/c4_p${p}_$i.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/c4_p${p}_$i.json')); print('res_prio=$p', 'csv us/iter', round(d['roofline']['avg_launch_us'],2), 'pm us/step', round(d['pm']['us_per_step'],3), d['checked'])"
done; done
N=2048 REPS=4 timeout -k 10 300 python tools/ab_probe.py "resident=1,res_prio=0" "resident=1,res_prio=1" > $O/ab_2048.log 2>&1; cat $O/ab_2048.log
N=1024 REPS=4 timeout -k 10 300 python tools/ab_probe.py "resident=1,res_prio=0" "resident=1,res_prio=1" > $O/ab_1024.log 2>&1; cat $O/ab_1024.log
timeout -k 10 600 python -m pytest tests/test_gpu_resident.py tests/test_gpu_pm_resident.py tests/test_gpu_resident_fuzz.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log

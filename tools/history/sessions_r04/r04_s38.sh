#!/bin/bash
# round 4, session 38: the master tile issues its first polls in front of its border stores; its wave 0 stores no border
set -o pipefail
O=gpurun_out/r4s38; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_resident.py tests/test_gpu_resident_fuzz.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
grep -q "rc=0" $O/pytest.log || exit 1
for m in 0 1; do
timeout -k 10 200 python tools/resident_timeline.py res_master=$m res_go_share=5 > $O/resident_timeline_2048_master$m.txt 2>&1; tail -9 $O/resident_timeline_2048_master$m.txt
done
N=2048 REPS=4 timeout -k 10 300 python tools/ab_probe.py "resident=1,res_master=0,res_go_share=0" "resident=1,res_master=1,res_go_share=0" "resident=1,res_master=1,res_go_share=5" > $O/ab_2048.log 2>&1; cat $O/ab_2048.log
N=1024 REPS=4 timeout -k 10 300 python tools/ab_probe.py "resident=1,res_master=0,res_go_share=0" "resident=1,res_master=1,res_go_share=0" "resident=1,res_master=1,res_go_share=5" > $O/ab_1024.log 2>&1; cat $O/ab_1024.log
for i in 1 2; do
  timeout -k 10 200 python bench.py --config C4 --no-cpu-baseline --no-phases --opt res_go_share=5 > $O/c4_$i.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/c4_$i.json')); print('csv us/iter', round(d['roofline']['avg_launch_us'],2), 'pm us/step', round(d['pm']['us_per_step'],3), d['checked'])"
done

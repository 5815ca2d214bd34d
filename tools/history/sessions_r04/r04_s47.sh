#!/bin/bash
# round 4, session 47: resident Perona-Malik without the workgroup-wide wait for the halo ring -- every wave gathers the cells its own band reads, polled early
set -o pipefail
O=gpurun_out/r4s47; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_pm_resident.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
grep -q "rc=0" $O/pytest.log || exit 1
B=tools/experiments/_libs/libchanvese_hip_r04_pm_upfront.so
for n in 2048 1024 512; do
N=$n REPS=4 OPTS=pm_kernel=4 timeout -k 10 300 python tools/pm_ab_libs.py $B chan_vese_amd/csrc/libchanvese_hip.so > $O/pm_ab_libs_$n.log 2>&1; cat $O/pm_ab_libs_$n.log
done
timeout -k 10 200 python tools/pm_resident_timeline.py > $O/pm_resident_timeline_2048.txt 2>&1; tail -18 $O/pm_resident_timeline_2048.txt
for i in 1 2; do
  timeout -k 10 200 python bench.py --config C4 --no-cpu-baseline --no-phases > $O/c4_$i.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/c4_$i.json')); print('csv us/iter', round(d['roofline']['avg_launch_us'],2), 'pm us/step', round(d['pm']['us_per_step'],3), d['checked'])"
done

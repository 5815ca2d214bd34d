#!/bin/bash
# round 4, session 59: what an iteration reads that does not depend on the region means is taken before the release is waited for
set -o pipefail
O=gpurun_out/r4s59; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_resident.py tests/test_gpu_resident_fuzz.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
grep -q "rc=0" $O/pytest.log || exit 1
B=tools/experiments/_libs/libchanvese_hip_r04_prereads0.so
for n in 2048 1024 512; do
N=$n REPS=5 OPTS=resident=1 timeout -k 10 300 python tools/ab_libs.py $B chan_vese_amd/csrc/libchanvese_hip.so > $O/ab_libs_$n.log 2>&1; cat $O/ab_libs_$n.log
done
timeout -k 10 200 python tools/resident_timeline.py > $O/resident_timeline_2048.txt 2>&1; tail -9 $O/resident_timeline_2048.txt | cut -c1-250

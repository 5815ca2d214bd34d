#!/bin/bash
# round 4, session 51: (a) resident CSV with 16-byte borders, three-piece arrival: tests + bench; (b) experiment: resident Perona-Malik publishing its border from LDS (full coalesced stores) instead of from registers
set -o pipefail
O=gpurun_out/r4s51; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_resident.py tests/test_gpu_resident_fuzz.py tests/test_gpu_pm_resident.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
grep -q "rc=0" $O/pytest.log || exit 1
B=tools/experiments/_libs/libchanvese_hip_r04_pmpoll0.so
for n in 2048 1024 512; do
N=$n REPS=4 OPTS=pm_kernel=4 timeout -k 10 300 python tools/pm_ab_libs.py $B chan_vese_amd/csrc/libchanvese_hip.so > $O/pm_ab_libs_$n.log 2>&1; cat $O/pm_ab_libs_$n.log
done

#!/bin/bash
# round 3, session 8: the resident kernel: first parity run, then its time against the per-launch flow
set -o pipefail
O=gpurun_out/r3s8; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_resident.py -m gpu -x -q --durations=5 > $O/pytest_resident.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest_resident.log; tail -30 $O/pytest_resident.log
[ $rc -eq 0 ] || exit 0
N=2048 REPS=3 STEPS=200 timeout -k 10 300 python tools/ab_probe.py "resident=0" "resident=1" > $O/ab2048.txt 2>&1; cat $O/ab2048.txt
N=1024 REPS=3 STEPS=200 timeout -k 10 300 python tools/ab_probe.py "resident=0" "resident=1" > $O/ab1024.txt 2>&1; cat $O/ab1024.txt
N=512 REPS=3 STEPS=200 timeout -k 10 300 python tools/ab_probe.py "resident=0" "resident=1" > $O/ab512.txt 2>&1; cat $O/ab512.txt
V=chan_vese_amd/csrc/variants; D=chan_vese_amd/csrc/libchanvese_hip.so
[ -f $V/res16/libchanvese_hip.so ] && { N=2048 REPS=3 STEPS=200 OPTS=resident=1 timeout -k 10 300 python tools/ab_libs.py $D $V/res16/libchanvese_hip.so > $O/ab_res16.txt 2>&1; cat $O/ab_res16.txt; }

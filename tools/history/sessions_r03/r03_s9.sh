#!/bin/bash
# round 3, session 9: resident kernel, arrival lines + master: parity, timeline, time against the per-launch flow
set -o pipefail
O=gpurun_out/r3s9; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_resident.py -m gpu -x -q > $O/pytest_resident.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest_resident.log; tail -25 $O/pytest_resident.log
[ $rc -eq 0 ] || exit 0
N=2048 timeout -k 5 120 python tools/resident_timeline.py > $O/tl2048.txt 2>&1; cat $O/tl2048.txt
N=512 timeout -k 5 120 python tools/resident_timeline.py > $O/tl512.txt 2>&1; tail -4 $O/tl512.txt
for n in 2048 1024 512; do N=$n REPS=3 STEPS=200 timeout -k 10 300 python tools/ab_probe.py "resident=0" "resident=1" > $O/ab$n.txt 2>&1; cat $O/ab$n.txt; done
V=chan_vese_amd/csrc/variants; D=chan_vese_amd/csrc/libchanvese_hip.so
[ -f $V/res12/libchanvese_hip.so ] && { CHANVESE_HIP_LIB=$V/res12/libchanvese_hip.so timeout -k 10 300 python -m pytest tests/test_gpu_resident.py -m gpu -x -q -k "small_shapes or 2048" > $O/pytest_res12.log 2>&1; tail -3 $O/pytest_res12.log; for n in 2048 1024; do CHANVESE_HIP_LIB=$V/res12/libchanvese_hip.so N=$n REPS=3 STEPS=200 timeout -k 10 300 python tools/ab_probe.py "resident=1" > $O/ab${n}_res12.txt 2>&1; cat $O/ab${n}_res12.txt; done; }

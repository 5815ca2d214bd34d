#!/bin/bash
# round 3, session 29: straight-line flavours of csv_resident_kernel: parity, then A/B against the generic march in one context
set -o pipefail
O=gpurun_out/r3s29; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_resident.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest.log; tail -6 $O/pytest.log
[ $rc -eq 0 ] || exit 1
N=2048 STEPS=1024 timeout -k 10 200 python tools/ab_probe.py "resident=1,res_straight=1" "resident=1,res_straight=0" > $O/ab2048.log 2>&1; cat $O/ab2048.log
N=1024 STEPS=1024 timeout -k 10 200 python tools/ab_probe.py "resident=1,res_straight=1" "resident=1,res_straight=0" > $O/ab1024.log 2>&1; cat $O/ab1024.log
N=512 STEPS=1024 timeout -k 10 200 python tools/ab_probe.py "resident=1,res_straight=1" "resident=1,res_straight=0" > $O/ab512.log 2>&1; cat $O/ab512.log

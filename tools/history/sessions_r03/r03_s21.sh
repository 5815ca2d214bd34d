#!/bin/bash
# round 3, session 21: resident Perona-Malik, separable row pass + straight-line bands of 2/4/8/16 rows + border from registers
set -o pipefail
O=gpurun_out/r3s21; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_pm_resident.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest.log; tail -30 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/pm_flows.py > $O/pm_ab.log 2>&1; cat $O/pm_ab.log
N=2048 timeout -k 10 200 python tools/pm_resident_timeline.py > $O/tl_2048.log 2>&1; tail -11 $O/tl_2048.log

#!/bin/bash
# round 3, session 28: resident Perona-Malik with 12 waves x 11 rows (three waves per SIMD) against 8 x 16
set -o pipefail
O=gpurun_out/r3s28; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_pm_resident.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest.log; tail -4 $O/pytest.log
[ $rc -eq 0 ] || exit 1
for args in "132 128 3 1 12" "132 256 5 2 12" "300 260 7 1 12" "700 384 9 2 12" "2048 2048 21 1 12" "2048 2048 40 2 12" "1000 1000 12 1 12"; do timeout -k 10 100 python tools/history/dbg_pm_resident.py $args 2>&1 | tail -3; done > $O/dbg12.log 2>&1; cat $O/dbg12.log | cut -c1-250
SIZES=1024,1536,2048 timeout -k 10 300 python tools/pm_flows.py > $O/w8.log 2>&1; cat $O/w8.log
SIZES=1024,1536,2048 OPTS=pm_res_waves=12 timeout -k 10 300 python tools/pm_flows.py > $O/w12.log 2>&1; cat $O/w12.log

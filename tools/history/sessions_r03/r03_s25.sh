#!/bin/bash
# round 3, session 25: PM tests after the auto rule; C4 evidence re-collected with the resident Perona-Malik kernel (bench, kernel trace, PMC)
set -o pipefail
O=gpurun_out/r3s25; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_pm_resident.py tests/test_gpu_parity.py -m gpu -x -q -k "pm or perona or launch_info" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest.log; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
SIZES=128,256,512,1024,1536,2048 timeout -k 10 300 python tools/pm_flows.py > $O/pm_flows.log 2>&1; cat $O/pm_flows.log
N=2048 timeout -k 10 200 python tools/pm_resident_timeline.py > $O/pm_timeline_2048.log 2>&1
bash tools/profile_round.sh r03_C4 "--config C4" > gpurun_out/r03_C4_summary.txt 2>&1; tail -14 gpurun_out/r03_C4_summary.txt | cut -c1-400

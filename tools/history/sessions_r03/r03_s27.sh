#!/bin/bash
# round 3, session 27: full GPU suite + smoke; C4 evidence with the tagged-border resident Perona-Malik kernel
set -o pipefail
O=gpurun_out/r3s27; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest.log; tail -10 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
SIZES=128,256,512,1024,1536,2048 timeout -k 10 300 python tools/pm_flows.py > $O/pm_flows.log 2>&1; cat $O/pm_flows.log
N=2048 timeout -k 10 200 python tools/pm_resident_timeline.py > $O/pm_timeline_2048.log 2>&1
bash tools/profile_round.sh r03_C4 "--config C4" > gpurun_out/r03_C4_summary.txt 2>&1; tail -3 gpurun_out/r03_C4_summary.txt | cut -c1-300

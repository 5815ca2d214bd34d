#!/bin/bash
# round 3, session 18: full GPU suite + smoke + C4 evidence re-collected (resident kernel: sums in the arrival lines, barrier-free master)
set -o pipefail
O=gpurun_out/r3s18; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest.log; tail -12 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 200 python tools/resident_timeline.py > $O/timeline.log 2>&1; tail -14 $O/timeline.log
bash tools/profile_round.sh r03_C4 "--config C4" > gpurun_out/r03_C4_summary.txt 2>&1; tail -12 gpurun_out/r03_C4_summary.txt

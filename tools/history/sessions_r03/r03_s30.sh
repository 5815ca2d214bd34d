#!/bin/bash
# round 3, session 30: full GPU suite + smoke; C4 evidence with the straight-line csv_resident_kernel<16>; timeline
set -o pipefail
O=gpurun_out/r3s30; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest.log; tail -10 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 200 python tools/resident_timeline.py > $O/timeline.log 2>&1; tail -13 $O/timeline.log
for n in 128 256 512 768 1024 1536 2048; do N=$n STEPS=512 REPS=3 timeout -k 10 100 python tools/ab_probe.py "resident=1" "resident=0" 2>&1 | tail -2; done > $O/resident_sizes.log; cat $O/resident_sizes.log
bash tools/profile_round.sh r03_C4 "--config C4" > gpurun_out/r03_C4_summary.txt 2>&1; tail -3 gpurun_out/r03_C4_summary.txt | cut -c1-200

#!/bin/bash
# round 4, session 42: the whole GPU suite with the new master, then the C4 evidence again (bench line, kernel stats, single-launch stats, counters)
set -o pipefail
O=gpurun_out/r4s42; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
grep -q "rc=0" $O/pytest.log || exit 1
bash tools/profile_round.sh r04_C4b "--config C4" > $O/profile_C4.log 2>&1; tail -30 $O/profile_C4.log

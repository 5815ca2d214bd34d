#!/bin/bash
# round 4, session 10: the suite with the co-residency rules; the batch probe again (auto must now match resident=0); profiles part 1
set -o pipefail
O=gpurun_out/r4s10; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
timeout -k 10 300 python tools/batch_probe.py > $O/batch_2048.log 2>&1; cat $O/batch_2048.log
bash tools/history/sessions_r04/r04_prof1.sh

#!/bin/bash
# round 3, session 24: full GPU suite + smoke with the resident Perona-Malik kernel as the default; short runs; C4 bench line
set -o pipefail
O=gpurun_out/r3s24; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=5 > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest.log; tail -12 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
for st in 1 2 4 8; do SIZES=64,128,512,2048 STEPS=$st REPS=3 timeout -k 10 200 python tools/pm_flows.py; done > $O/pm_short.log 2>&1; cat $O/pm_short.log
timeout -k 10 400 python bench.py --config C4 > $O/bench_C4.json 2> $O/bench_C4.err; python -c "
import json; d=json.load(open('$O/bench_C4.json')); print(d['value'], d['checked']); print(d.get('phases')); print({k:v for k,v in d.items() if 'pm' in k.lower()})"

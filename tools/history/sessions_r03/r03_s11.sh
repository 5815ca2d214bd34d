#!/bin/bash
# round 3, session 11: the whole GPU suite with resident mode as the automatic flow for planes that fit; smoke; C4 bench line
set -o pipefail
O=gpurun_out/r3s11; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -25 $O/pytest.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log
timeout -k 10 400 python bench.py --config C4 > $O/bench_C4.json 2> $O/bench_C4.err; python - <<PY
import json
try:
    d=json.load(open("$O/bench_C4.json")); print("C4", d["value"], d["roofline"]["avg_launch_us"], d["roofline"]["frac"], d["roofline"]["frac_wall"], d["roofline"]["kernel"], d["checked"], d["pm"]["us_per_step"])
except Exception as e: print("C4 failed", e, open("$O/bench_C4.err").read()[-1500:])
PY

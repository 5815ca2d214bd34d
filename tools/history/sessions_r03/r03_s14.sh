#!/bin/bash
# round 3, session 14: full GPU suite + smoke + the driver's command + C3 line after the 3-channel default changed
set -o pipefail
O=gpurun_out/r3s14; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=8 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -16 $O/pytest.log
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
timeout -k 10 400 python bench.py --config C3 > $O/bench_C3.json 2> $O/bench_C3.err
python - <<PY
import json
for f in ("driver","C3"):
    try:
        d=json.load(open("$O/bench_%s.json"%f)); print(f, d["value"], d["roofline"]["kernel"], round(d["roofline"]["avg_launch_us"],2), round(d["roofline"]["frac"],4), round(d["roofline"]["frac_wall"],4), d["checked"], (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e: print(f,"failed",e)
PY

#!/bin/bash
# round 3, session 35: final check: full GPU suite (incl. the seeded fuzz test) + smoke + the driver's command + C3 / C4 / C5 lines
set -o pipefail
O=gpurun_out/r3s35; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=6 > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest.log; tail -12 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err
timeout -k 10 400 python bench.py > $O/bench_C2.json 2> $O/bench_C2.err
timeout -k 10 400 python bench.py --config C4 > $O/bench_C4.json 2> $O/bench_C4.err
python - <<PY
import json
for f in ("driver","C2","C4"):
    try:
        d=json.load(open("$O/bench_%s.json"%f)); r=d["roofline"]
        print(f, round(d["value"],1), r["kernel"], round(r["avg_launch_us"],2), round(r["frac"],4), round(r["frac_wall"],4), d["checked"], (d.get("cpu_baseline") or {}).get("value"), (d.get("pm") or {}).get("us_per_step"))
    except Exception as e: print(f,"failed",e)
PY

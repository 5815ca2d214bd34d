#!/bin/bash
# round 3, session 36: the C3 and C5 lines with the final build (same box, one call)
set -o pipefail
O=gpurun_out/r3s36; mkdir -p $O
timeout -k 10 400 python bench.py --config C3 > $O/bench_C3.json 2> $O/bench_C3.err
timeout -k 10 500 python bench.py --config C5 > $O/bench_C5.json 2> $O/bench_C5.err
python - <<PY
import json
for f in ("C3","C5"):
    try:
        d=json.load(open("$O/bench_%s.json"%f)); r=d["roofline"]
        print(f, round(d["value"],1), r["kernel"], round(r["avg_launch_us"],2), round(r["frac"],4), round(r["frac_wall"],4), d["checked"], (d.get("cpu_baseline") or {}).get("value"))
    except Exception as e: print(f,"failed",e)
PY

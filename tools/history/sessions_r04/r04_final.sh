#!/bin/bash
# round 4: last session -- the whole GPU suite, smoke(), the driver's command and the default bench lines of every config on one box
set -o pipefail
O=gpurun_out/r04_final; mkdir -p $O
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver_cmd.json 2> $O/bench_driver_cmd.err
timeout -k 10 300 python bench.py > $O/bench_default.json 2> $O/bench_default.err
for cfg in C3 C4 C5 near C2-f32 C3-f32; do timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline > $O/bench_$cfg.json 2> $O/bench_$cfg.err; done
python - $O <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/bench_*.json")):
    try:
        d=json.load(open(f)); r=d["roofline"]
        print(f.split("/")[-1], "us", round(r["avg_launch_us"],2), r["bound"], "frac", r["frac"] and round(r["frac"],4), "wall", r.get("frac_wall") and round(r["frac_wall"],4), "value", round(d["value"]), "checked", d["checked"], (d.get("pm") or {}).get("us_per_step"))
    except Exception as e: print(f, "ERR", e)
PY

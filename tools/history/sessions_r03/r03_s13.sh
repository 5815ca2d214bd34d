#!/bin/bash
# round 3, session 13: 3-channel kernels, 1-pixel vs 2-pixel, decisive A/B (one context, alternating; then bench lines)
set -o pipefail
O=gpurun_out/r3s13; mkdir -p $O
C=3 N=4096 REPS=5 STEPS=300 timeout -k 10 400 python tools/ab_probe.py "kernel=2" "kernel=3" "kernel=3,wave_cskew=0" "kernel=3,wave_pol=1" > $O/ab_c3.txt 2>&1; cat $O/ab_c3.txt
timeout -k 10 300 python bench.py --config C3 --no-cpu-baseline --no-phases > $O/bench_c3_k2.json 2>$O/e1; timeout -k 10 300 python bench.py --config C3 --no-cpu-baseline --no-phases --opt kernel=3 > $O/bench_c3_k3.json 2>$O/e2
timeout -k 10 300 python bench.py --config C3 --no-cpu-baseline --no-phases > $O/bench_c3_k2b.json 2>$O/e3; timeout -k 10 300 python bench.py --config C3 --no-cpu-baseline --no-phases --opt kernel=3 --opt wave_cskew=0 > $O/bench_c3_k3b.json 2>$O/e4
python - <<PY
import json
for f in ("k2","k3","k2b","k3b"):
    try:
        d=json.load(open("$O/bench_c3_%s.json"%f)); print(f, d["roofline"]["kernel"], round(d["roofline"]["avg_launch_us"],2), round(d["roofline"]["frac"],4), d["checked"])
    except Exception as e: print(f,"failed",e)
PY

#!/bin/bash
# round 4, session 18: class skew 0.425 as the 3-channel default: bench line on another box, A/B against equal strips, the 3-channel tests
set -o pipefail
O=gpurun_out/r4s18; mkdir -p $O
timeout -k 10 300 python bench.py --config C3 --no-cpu-baseline > $O/bench_C3.json 2> $O/bench_C3.err; python -c "
import json; d=json.load(open('$O/bench_C3.json')); r=d['roofline']; print('C3', round(r['avg_launch_us'],2), round(r['frac'],4), round(r['frac_wall'],4), round(d['value']), d['checked'])"
C=3 REPS=4 STEPS=304 timeout -k 10 600 python tools/ab_probe.py "wave_cskew=0" "wave_cskew=500" "wave_cskew=0" "wave_cskew=500" > $O/ab_c3.log 2>&1; cat $O/ab_c3.log
timeout -k 10 300 python bench.py --config C3-f32 --no-cpu-baseline > $O/bench_C3f32.json 2> $O/bench_C3f32.err; python -c "
import json; d=json.load(open('$O/bench_C3f32.json')); r=d['roofline']; print('C3-f32', round(r['avg_launch_us'],2), round(r['frac'],4), round(d['value']), d['checked'])"
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "three_channel or config3 or C3 or readme or state32" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log

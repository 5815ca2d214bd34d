#!/bin/bash
# round 3, session 2: extended hazard probe, the new 3-channel 2-pixel flavour (parity, then A/B against the 1-pixel kernel),
# bench lines of every config with the result check
set -o pipefail
O=gpurun_out/r3s2; mkdir -p $O


# (the probe ran in the first attempt of this session: gpurun_out/r3s2/hazard_probe.txt)
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "three_channel_two_pixel or launch_info or three_channel_flavours or beyond_the_cache or rccl or adjudicated" --durations=10 > $O/pytest_new.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest_new.log; tail -5 $O/pytest_new.log
# (a failing test does not stop the measurements below; its log is kept)
C=3 timeout -k 10 300 python tools/ab_probe.py "kernel=2" "kernel=3" "kernel=3,lut=0" "kernel=3,wave_pol=1" "kernel=3,wave_cskew=0" > $O/ab_c3.log 2>&1; cat $O/ab_c3.log
C=3 N=2048 timeout -k 10 300 python tools/ab_probe.py "kernel=2" "kernel=3" "kernel=3,lut=0" > $O/ab_c3_2048.log 2>&1; cat $O/ab_c3_2048.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_c2_driver.json 2> $O/bench_c2_driver.err; tail -c 1500 $O/bench_c2_driver.json
for cfg in C3 C4 C5; do timeout -k 10 400 python bench.py --config $cfg --no-cpu-baseline > $O/bench_$cfg.json 2> $O/bench_$cfg.err; python - <<PY
import json
try:
    d=json.load(open("$O/bench_$cfg.json")); print("$cfg", d["value"], d["roofline"]["frac"], d["roofline"]["frac_wall"], d["roofline"]["kernel"], d["checked"], d.get("pm",{}).get("roofline",{}).get("kernel"))
except Exception as e: print("$cfg failed", e, open("$O/bench_$cfg.err").read()[-800:])
PY
done

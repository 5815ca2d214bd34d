#!/bin/bash
mkdir -p gpurun_out/aux
: > gpurun_out/aux/pol4.log
for hw in "4352 4352" "4320 7680" "8192 8192"; do set -- $hw
  H=$1 W=$2 REPS=3 STEPS=48 python tools/ab_probe.py "kernel=3,wave_pol=0" "kernel=3,wave_pol=2" >> gpurun_out/aux/pol4.log 2>&1
done
cat gpurun_out/aux/pol4.log
for p in 0 2 0 2; do python bench.py --config C5 --no-cpu-baseline --steps 200 --warmup 40 --opt wave_pol=$p > gpurun_out/aux/c5.json 2>gpurun_out/aux/e.log; python -c "import json;d=json.load(open('gpurun_out/aux/c5.json'));print('C5 wave_pol=$p', round(d['value']))"; done

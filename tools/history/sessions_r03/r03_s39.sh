#!/bin/bash
# round 3, session 39: wave_sync auto (0 for three channels): the flavour tests at 4096^2, the 3-channel parity tests, the C3 line
set -o pipefail
O=gpurun_out/r3s39; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py -m gpu -x -q -k "flavours or three_channel or config3 or C3 or golden" > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest.log; tail -5 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 400 python bench.py --config C3 > $O/bench_C3.json 2> $O/bench_C3.err
python -c "
import json; d=json.load(open('$O/bench_C3.json')); r=d['roofline']; print(d['value'], r['kernel'], r['avg_launch_us'], r['frac'], r['frac_wall'], d['checked'])"
C=3 N=4096 REPS=3 STEPS=112 timeout -k 10 400 python tools/ab_probe.py "kernel=3" "kernel=3,wave_sync=1" "kernel=3,wave_sync=0" > $O/ab.log 2>&1; cat $O/ab.log

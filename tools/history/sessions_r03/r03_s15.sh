#!/bin/bash
# round 3, session 15: resident kernel with arrival lines that carry the fixed-point sums (no atomics, arrival ahead of the border stores)
set -o pipefail
O=gpurun_out/r3s15; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_resident.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest.log; tail -6 $O/pytest.log
[ $rc -eq 0 ] || exit 1
N=2048 timeout -k 10 200 python tools/ab_probe.py "resident=1" "resident=0" > $O/ab.log 2>&1; cat $O/ab.log
timeout -k 10 200 python tools/resident_timeline.py > $O/timeline.log 2>&1; tail -25 $O/timeline.log
timeout -k 10 400 python bench.py --config C4 > $O/bench_C4.json 2> $O/bench_C4.err; python -c "
import json; d=json.load(open('$O/bench_C4.json')); print(d['value'], d['roofline'], d['checked'])"

#!/bin/bash
# round 4, session 8: the declared FP32-state mode (tests, first bench lines), then the whole suite
set -o pipefail
O=gpurun_out/r4s8; mkdir -p $O
timeout -k 10 900 python -m pytest tests/test_gpu_state32.py -x -q -m gpu -s > $O/pytest_state32.log 2>&1; echo "rc=$?" >> $O/pytest_state32.log; tail -12 $O/pytest_state32.log
timeout -k 10 900 python -m pytest tests/test_gpu_fullsize.py -x -q -m gpu -s -k state32 > $O/pytest_state32_full.log 2>&1; echo "rc=$?" >> $O/pytest_state32_full.log; grep "state 32\|passed\|failed\|rc=" $O/pytest_state32_full.log
for cfg in C2-f32 C3-f32 C2 C3; do
timeout -k 10 300 python bench.py --config $cfg --no-cpu-baseline > $O/bench_$cfg.json 2> $O/bench_$cfg.err; python - $O/bench_$cfg.json <<'PY'
import json,sys
d=json.load(open(sys.argv[1])); r=d["roofline"]
print(d["config"]["workload"][:20], d["dtype"], "us/launch", round(r["avg_launch_us"],2), "frac", round(r["frac"],4), "Mpx-it/s", round(d["value"]), r["kernel"], "checked", d["checked"])
PY
done
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -8 $O/pytest.log

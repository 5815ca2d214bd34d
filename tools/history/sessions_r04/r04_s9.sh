#!/bin/bash
# round 4, session 9: batch of eight 2048^2 images, resident vs per-launch flows; the suite with the automatic cache policy; then profiles part 1
set -o pipefail
O=gpurun_out/r4s9; mkdir -p $O
timeout -k 10 300 python tools/batch_probe.py > $O/batch_2048.log 2>&1; cat $O/batch_2048.log
N=1024 timeout -k 10 300 python tools/batch_probe.py > $O/batch_1024.log 2>&1; cat $O/batch_1024.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log
timeout -k 10 300 python bench.py --config C5 --no-cpu-baseline > $O/bench_C5.json 2> $O/bench_C5.err; cut -c1-300 $O/bench_C5.json

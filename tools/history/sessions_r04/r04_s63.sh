#!/bin/bash
# round 4, session 63: the chunk-aware rule for a batch (large planes in long chunks take the resident flow): tests, batch probe "auto" lines
set -o pipefail
O=gpurun_out/r4s63; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_resident.py tests/test_gpu_fullsize.py -x -q -m gpu -k "batch or resident or config5 or interleaved" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log
N=2048 REPS=2 timeout -k 10 400 python tools/batch_probe.py > $O/batch_2048.log 2>&1; cat $O/batch_2048.log
N=1536 REPS=2 timeout -k 10 400 python tools/batch_probe.py > $O/batch_1536.log 2>&1; cat $O/batch_1536.log

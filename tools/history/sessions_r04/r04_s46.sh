#!/bin/bash
# round 4, session 46: the new release-line test; the resident fuzz for a few minutes with the new master
set -o pipefail
O=gpurun_out/r4s46; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_resident.py -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -4 $O/pytest.log
timeout -k 10 400 python tools/fuzz_resident.py > $O/fuzz.log 2>&1; echo "fuzz rc=$?" >> $O/fuzz.log; tail -5 $O/fuzz.log

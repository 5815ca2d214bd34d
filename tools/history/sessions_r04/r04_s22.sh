#!/bin/bash
# round 4, session 22: after resetting the per-run automatic choices at run reset: the tests that exercise them
set -o pipefail
O=gpurun_out/r4s22; mkdir -p $O
timeout -k 10 900 python -m pytest tests -x -q -m gpu -k "batch or automatic_flow or interleaved or resident or launch_info or bench" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -5 $O/pytest.log

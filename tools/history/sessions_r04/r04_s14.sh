#!/bin/bash
# round 4, session 14: the sustained all-near regime in every FAST flow (new parity test) + the CLI --state test
set -o pipefail
O=gpurun_out/r4s14; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_parity.py tests/test_cli.py -x -q -m gpu -k "all_near_regime or state_32" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -12 $O/pytest.log

#!/bin/bash
# round 3, session 19: first run of the resident Perona-Malik kernel: parity, then timing against the 2-step per-launch kernel
set -o pipefail
O=gpurun_out/r3s19; mkdir -p $O
timeout -k 10 600 python -m pytest tests/test_gpu_pm_resident.py -m gpu -x -q > $O/pytest.log 2>&1; rc=$?; echo "pytest rc=$rc" >> $O/pytest.log; tail -30 $O/pytest.log
[ $rc -eq 0 ] || exit 1
timeout -k 10 300 python tools/pm_flows.py > $O/pm_ab.log 2>&1; cat $O/pm_ab.log

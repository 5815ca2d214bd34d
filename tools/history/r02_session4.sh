#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s4
timeout -k 10 300 tools/mall_split_probe > gpurun_out/s4/mall_split.log 2>&1; echo "mall rc=$?"; cat gpurun_out/s4/mall_split.log
python -m pytest tests -m gpu -q -x > gpurun_out/s4/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/s4/pytest_all.log

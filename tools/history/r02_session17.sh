#!/bin/bash
mkdir -p gpurun_out/s17
python -m pytest tests -m gpu -q -x > gpurun_out/s17/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/s17/pytest_all.log
python bench.py --no-cpu-baseline > gpurun_out/s17/bench_default.json 2>&1; tail -c 700 gpurun_out/s17/bench_default.json

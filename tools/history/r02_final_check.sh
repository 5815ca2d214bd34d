#!/bin/bash
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -q -x > gpurun_out/final/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/final/pytest_all.log
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/final/smoke.log 2>&1; echo "smoke rc=$?"; tail -1 gpurun_out/final/smoke.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final/bench_driver.json 2>gpurun_out/final/bench_driver.err; echo "bench rc=$?"; tail -c 1200 gpurun_out/final/bench_driver.json

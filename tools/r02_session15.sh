#!/bin/bash
mkdir -p gpurun_out/s15
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_parity.py tests/test_golden.py -m gpu -q -x -k "three_channel or 3ch or golden or config3" > gpurun_out/s15/pytest_sub.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/s15/pytest_sub.log
C=3 python tools/ab_probe.py kernel=2 kernel=3 "kernel=3,wave_cskew=0" "kernel=3,wave_cskew=300" > gpurun_out/s15/ab_c3.log 2>&1; cat gpurun_out/s15/ab_c3.log
C=3 N=2048 python tools/ab_probe.py kernel=2 kernel=3 > gpurun_out/s15/ab_c3_2048.log 2>&1; cat gpurun_out/s15/ab_c3_2048.log

#!/bin/bash
mkdir -p gpurun_out/pmk2b
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "perona or pm_then or pm_flavours or config4" > gpurun_out/pmk2b/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/pmk2b/pytest.log
python tools/pm_ab.py pm_kernel=3 "pm_kernel=3,pm_strip_rows=24" "pm_kernel=3,pm_strip_rows=32" "pm_kernel=3,pm_strip_rows=40" "pm_kernel=3,pm_strip_rows=48" "pm_kernel=3,pm_strip_rows=64" pm_kernel=1 > gpurun_out/pmk2b/pm2048.log 2>&1; cat gpurun_out/pmk2b/pm2048.log
N=4096 python tools/pm_ab.py pm_kernel=3 "pm_kernel=3,pm_strip_rows=64" "pm_kernel=3,pm_strip_rows=104" "pm_kernel=3,pm_strip_rows=128" > gpurun_out/pmk2b/pm4096.log 2>&1; cat gpurun_out/pmk2b/pm4096.log
N=1024 python tools/pm_ab.py pm_kernel=3 "pm_kernel=3,pm_strip_rows=16" "pm_kernel=3,pm_strip_rows=32" > gpurun_out/pmk2b/pm1024.log 2>&1; cat gpurun_out/pmk2b/pm1024.log

#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s7
python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "perona or pm_then" > gpurun_out/s7/pytest_pm.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/s7/pytest_pm.log
python tools/pm_ab.py pm_kernel=1 pm_kernel=3 "pm_kernel=3,pm_strip_rows=16" "pm_kernel=3,pm_strip_rows=32" "pm_kernel=3,pm_strip_rows=40" "pm_kernel=3,pm_strip_rows=48" "pm_kernel=3,pm_strip_rows=64" "pm_kernel=1,graph=0" > gpurun_out/s7/pm_ab.log 2>&1; echo "pmab rc=$?"; cat gpurun_out/s7/pm_ab.log
N=4096 python tools/pm_ab.py pm_kernel=2 pm_kernel=3 "pm_kernel=3,pm_strip_rows=48" "pm_kernel=3,pm_strip_rows=64" > gpurun_out/s7/pm_ab4096.log 2>&1; echo "pmab4096 rc=$?"; cat gpurun_out/s7/pm_ab4096.log
python tools/ab_probe.py "wave_cskew=0" "wave_cskew=300" "wave_cskew=400" "wave_cskew=500" "wave_cskew=600" "wave_cskew=750" "wave_cskew=900" "wave_cskew=400,wave_prio=2" "wave_cskew=600,wave_prio=2" > gpurun_out/s7/ab_skew.log 2>&1; echo "abs rc=$?"; cat gpurun_out/s7/ab_skew.log
python tools/ab_probe.py "strip_rows=0" "strip_rows=56" "strip_rows=62" "strip_rows=68" "strip_rows=76" "strip_rows=92" "strip_rows=68,wave_cskew=200" "strip_rows=36" "strip_rows=40" > gpurun_out/s7/ab_strip.log 2>&1; echo "abst rc=$?"; cat gpurun_out/s7/ab_strip.log
N=2048 python tools/ab_probe.py strip_rows=0 strip_rows=18 strip_rows=20 strip_rows=22 "strip_rows=20,wave_cskew=300" "strip_rows=0,wave_cskew=300" "strip_rows=20,wave_cls=0" > gpurun_out/s7/ab_2048.log 2>&1; echo "ab2048 rc=$?"; cat gpurun_out/s7/ab_2048.log

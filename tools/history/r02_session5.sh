#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s5
python -m pytest tests -m gpu -q > gpurun_out/s5/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -8 gpurun_out/s5/pytest_all.log
V=$PWD/chan_vese_amd/csrc/variants
python tools/ab_probe.py wave_cskew=0 wave_cskew=160 "wave_cskew=0,wave_prio=0" "wave_cskew=160,wave_prio=2" "wave_cskew=160,wave_prio=3" "wave_cskew=0,wave_sync=0" > gpurun_out/s5/ab2.log 2>&1; echo "ab2 rc=$?"; cat gpurun_out/s5/ab2.log
CHANVESE_HIP_LIB=$V/newton/libchanvese_hip.so python tools/ab_probe.py wave_cskew=0 wave_cskew=160 > gpurun_out/s5/ab_newton.log 2>&1; echo "abn rc=$?"; cat gpurun_out/s5/ab_newton.log
CHANVESE_HIP_LIB=$V/far4/libchanvese_hip.so python tools/ab_probe.py "wave_cskew=0,far_terms=4" "wave_cskew=160,far_terms=4" > gpurun_out/s5/ab_far4.log 2>&1; echo "abf rc=$?"; cat gpurun_out/s5/ab_far4.log
KERNEL=3 ITERS=40 SAVE=gpurun_out/s5/timeline.npz python tools/wave_timeline.py wave_cskew=160 > gpurun_out/s5/timeline.log 2>&1; echo "tl rc=$?"; head -8 gpurun_out/s5/timeline.log
python bench.py --config C4 --no-cpu-baseline > gpurun_out/s5/bench_C4.json 2>&1; echo "C4 rc=$?"; tail -c 900 gpurun_out/s5/bench_C4.json

#!/bin/bash
mkdir -p gpurun_out/s13
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py tests/test_cli.py -m gpu -q -x -k "two_pixel or kernel_variants or config2 or config1 or stop_rule or video or small_shapes" > gpurun_out/s13/pytest_sub.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s13/pytest_sub.log
python tools/ab_probe.py wave_early=1 wave_early=0 > gpurun_out/s13/ab_early.log 2>&1; cat gpurun_out/s13/ab_early.log
CHANVESE_HIP_LIB=$PWD/chan_vese_amd/csrc/variants/freq3/libchanvese_hip.so python tools/ab_probe.py wave_early=1 wave_early=0 > gpurun_out/s13/ab_early_freq3.log 2>&1; cat gpurun_out/s13/ab_early_freq3.log
N=2048 python tools/ab_probe.py wave_early=1 wave_early=0 > gpurun_out/s13/ab_early2048.log 2>&1; cat gpurun_out/s13/ab_early2048.log
KERNEL=3 ITERS=40 SAVE=gpurun_out/s13/timeline.npz python tools/wave_timeline.py > gpurun_out/s13/timeline.log 2>&1; head -4 gpurun_out/s13/timeline.log

#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s3
python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -k "stop or continu or interleav or two_pixel or config1 or golden or unlimited or nondefault" > gpurun_out/s3/pytest_sub.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s3/pytest_sub.log
python tools/ab_probe.py wave_cskew=0 wave_cskew=100 wave_cskew=130 wave_cskew=160 wave_cskew=200 wave_cskew=250 wave_cskew=300 "wave_cskew=0,chain=0" "wave_cskew=130,chain=0" "wave_cskew=200,chain=0" > gpurun_out/s3/ab1.log 2>&1; echo "ab1 rc=$?"; cat gpurun_out/s3/ab1.log
python bench.py --no-cpu-baseline > gpurun_out/s3/bench_default.json 2>gpurun_out/s3/bench_default.err; echo "bench rc=$?"; tail -c 1500 gpurun_out/s3/bench_default.json
python bench.py --no-cpu-baseline --opt wave_cskew=130 > gpurun_out/s3/bench_s130.json 2>&1; echo "bench130 rc=$?"; tail -c 1500 gpurun_out/s3/bench_s130.json
KERNEL=3 ITERS=40 SAVE=gpurun_out/s3/timeline_chain.npz python tools/wave_timeline.py wave_cskew=130 > gpurun_out/s3/timeline_chain.log 2>&1; echo "tl rc=$?"; cat gpurun_out/s3/timeline_chain.log | head -8

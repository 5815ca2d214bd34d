#!/bin/bash
# round 2, GPU session 1: new tests, the driver's bench command, the config legs, phases, timeline
set -o pipefail
mkdir -p gpurun_out/s1
python -m pytest tests/test_gpu_fullsize.py tests/test_cli.py -m gpu -x -q > gpurun_out/s1/pytest_new.log 2>&1; echo "pytest rc=$?" | tee -a gpurun_out/s1/pytest_new.log
tail -5 gpurun_out/s1/pytest_new.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s1/bench_driver.json 2> gpurun_out/s1/bench_driver.err; echo "driver-cmd rc=$?"
python bench.py > gpurun_out/s1/bench_default.json 2> gpurun_out/s1/bench_default.err; echo "default rc=$?"
python bench.py --config C3 --no-cpu-baseline > gpurun_out/s1/bench_C3.json 2> gpurun_out/s1/bench_C3.err; echo "C3 rc=$?"
python bench.py --config C4 --no-cpu-baseline > gpurun_out/s1/bench_C4.json 2> gpurun_out/s1/bench_C4.err; echo "C4 rc=$?"
python bench.py --config C5-image --no-cpu-baseline > gpurun_out/s1/bench_C5i.json 2> gpurun_out/s1/bench_C5i.err; echo "C5i rc=$?"
python bench.py --config C4-image --no-cpu-baseline > gpurun_out/s1/bench_C4i.json 2> gpurun_out/s1/bench_C4i.err; echo "C4i rc=$?"
python bench.py --config C5 --no-cpu-baseline --steps 100 --warmup 20 > gpurun_out/s1/bench_C5.json 2> gpurun_out/s1/bench_C5.err; echo "C5 rc=$?"
python tools/phase_probe.py > gpurun_out/s1/phase.log 2>&1; echo "phase rc=$?"
KERNEL=3 ITERS=40 SAVE=gpurun_out/s1/timeline_4096.npz python tools/wave_timeline.py > gpurun_out/s1/timeline.log 2>&1; echo "timeline rc=$?"
KERNEL=3 ITERS=3 SAVE=gpurun_out/s1/timeline_4096_it3.npz python tools/wave_timeline.py > gpurun_out/s1/timeline_it3.log 2>&1; echo "timeline3 rc=$?"
cat gpurun_out/s1/bench_driver.json gpurun_out/s1/bench_default.json

#!/bin/bash
# diagnostic: the parity subset that exercises every step-kernel variant, then the default bench
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py tests/test_golden.py -m gpu -x -q -k "kernel_variants or (small_shapes and 2) or three_channel or fixture or config1 or nondefault or stop" > gpurun_out/pytest_quick.log 2>&1
rc=$?
tail -3 gpurun_out/pytest_quick.log
[ $rc -eq 0 ] || exit $rc
for rep in 1 2; do
timeout -k 10 200 python bench.py --steps 300 --warmup 20 --no-cpu-baseline "$@" 2> gpurun_out/b.err | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bench', round(d['roofline']['avg_launch_us'],2), round(d['roofline']['frac'],3))"
done

#!/bin/bash
# round 4, session 11: profiles part 2 (C5, near, FP32-state lines), the C4 line with its valu_fp64 roofline filled from the committed counters
set -o pipefail
O=gpurun_out/r4s11; mkdir -p $O
timeout -k 10 300 python bench.py --config C4 > $O/bench_C4.json 2> $O/bench_C4.err; cut -c1-200 $O/bench_C4.json
bash tools/history/sessions_r04/r04_prof2.sh
timeout -k 10 600 python -m pytest tests/test_gpu_state32.py tests/test_gpu_resident.py -x -q -m gpu > $O/pytest_part.log 2>&1; echo "rc=$?" >> $O/pytest_part.log; tail -4 $O/pytest_part.log

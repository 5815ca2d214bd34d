#!/bin/bash
# round 3, session 33: randomised cross-check of the resident kernels against the per-launch flows
set -o pipefail
O=gpurun_out/r3s33; mkdir -p $O
CASES=80 SEED=7 timeout -k 10 900 python tools/fuzz_resident.py > $O/fuzz.log 2>&1; echo "rc=$?" >> $O/fuzz.log; tail -12 $O/fuzz.log | cut -c1-260

#!/bin/bash
# round 4, session 5: which of the round's changes to the 2-pixel kernel costs 3 us at 4096^2 (session 4: 61.3 vs 58.0-58.7 for the round-3 kernel)?
# orig | current (SDWA + integer LDS addresses + one min |u| per lane) | without SDWA | without integer LDS addressing | without both
set -o pipefail
O=gpurun_out/r4s5; mkdir -p $O
L=chan_vese_amd/csrc; V=$L/variants
REPS=3 timeout -k 10 400 python tools/ab_libs.py $V/orig/libchanvese_hip.so $L/libchanvese_hip.so $V/nosdwa/libchanvese_hip.so $V/nointlds/libchanvese_hip.so $V/nosdwa_nointlds/libchanvese_hip.so > $O/ab_c1.log 2>&1; cat $O/ab_c1.log
REPS=3 timeout -k 10 400 python tools/ab_libs.py $V/nosdwa_nointlds/libchanvese_hip.so $V/nointlds/libchanvese_hip.so $V/nosdwa/libchanvese_hip.so $L/libchanvese_hip.so $V/orig/libchanvese_hip.so > $O/ab_c1_rev.log 2>&1; cat $O/ab_c1_rev.log
C=3 REPS=3 timeout -k 10 400 python tools/ab_libs.py $V/orig/libchanvese_hip.so $L/libchanvese_hip.so $V/nosdwa/libchanvese_hip.so $V/nointlds/libchanvese_hip.so $V/nosdwa_nointlds/libchanvese_hip.so > $O/ab_c3.log 2>&1; cat $O/ab_c3.log
timeout -k 10 600 python -m pytest tests/test_gpu_readme_examples.py tests/test_gpu_fullsize.py -x -q -m gpu -k "readme or fixture or 500_iterations or batch_of_eight" > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log

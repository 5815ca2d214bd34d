#!/bin/bash
# round 4, session 7: after pruning (strip kernel, 2-pixel Perona-Malik kernel, quadratic 3-channel term, 4-waves/SIMD flavour, A/B macros):
# the GPU suite; and what in the round's changes to csv_resident_kernel costs 0.2-0.5 us at 2048^2 (near-band copy? SDWA / integer LDS?)
set -o pipefail
O=gpurun_out/r4s7; mkdir -p $O
L=chan_vese_amd/csrc; V=$L/variants
N=2048 OPTS=resident=1 REPS=4 timeout -k 10 300 python tools/ab_libs.py $V/orig/libchanvese_hip.so $L/libchanvese_hip.so $V/res_nonearband/libchanvese_hip.so $V/res_nosdwa_nointlds/libchanvese_hip.so $V/res_plain/libchanvese_hip.so > $O/ab_2048_resident.log 2>&1; cat $O/ab_2048_resident.log
N=2048 OPTS=resident=1 REPS=4 timeout -k 10 300 python tools/ab_libs.py $V/res_plain/libchanvese_hip.so $V/res_nosdwa_nointlds/libchanvese_hip.so $V/res_nonearband/libchanvese_hip.so $L/libchanvese_hip.so $V/orig/libchanvese_hip.so > $O/ab_2048_resident_rev.log 2>&1; cat $O/ab_2048_resident_rev.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log

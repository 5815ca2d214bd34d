#!/bin/bash
# round 4, session 6: four near masks again + per-strip near-form march + SDWA table addresses + integer LDS addressing, against the round-3
# kernel and against the same build without SDWA / integer LDS addressing, both orders; resident 2048^2; then the whole GPU suite
set -o pipefail
O=gpurun_out/r4s6; mkdir -p $O
L=chan_vese_amd/csrc; V=$L/variants
REPS=3 timeout -k 10 400 python tools/ab_libs.py $V/orig/libchanvese_hip.so $L/libchanvese_hip.so $V/nosdwa/libchanvese_hip.so $V/nointlds/libchanvese_hip.so $V/nosdwa_nointlds/libchanvese_hip.so > $O/ab_c1.log 2>&1; cat $O/ab_c1.log
REPS=3 timeout -k 10 400 python tools/ab_libs.py $V/nosdwa_nointlds/libchanvese_hip.so $V/nointlds/libchanvese_hip.so $V/nosdwa/libchanvese_hip.so $L/libchanvese_hip.so $V/orig/libchanvese_hip.so > $O/ab_c1_rev.log 2>&1; cat $O/ab_c1_rev.log
C=3 REPS=3 timeout -k 10 400 python tools/ab_libs.py $V/orig/libchanvese_hip.so $L/libchanvese_hip.so $V/nosdwa/libchanvese_hip.so $V/nointlds/libchanvese_hip.so $V/nosdwa_nointlds/libchanvese_hip.so > $O/ab_c3.log 2>&1; cat $O/ab_c3.log
C=3 REPS=3 timeout -k 10 400 python tools/ab_libs.py $V/nosdwa_nointlds/libchanvese_hip.so $V/nointlds/libchanvese_hip.so $V/nosdwa/libchanvese_hip.so $L/libchanvese_hip.so $V/orig/libchanvese_hip.so > $O/ab_c3_rev.log 2>&1; cat $O/ab_c3_rev.log
N=2048 OPTS=resident=1 timeout -k 10 300 python tools/ab_libs.py $L/libchanvese_hip.so $V/orig/libchanvese_hip.so > $O/ab_2048_resident.log 2>&1; cat $O/ab_2048_resident.log
N=2048 OPTS=resident=1 timeout -k 10 300 python tools/ab_libs.py $V/orig/libchanvese_hip.so $L/libchanvese_hip.so >> $O/ab_2048_resident.log 2>&1; tail -2 $O/ab_2048_resident.log
RESIDENT=0 timeout -k 10 400 python tools/near_regime_probe.py > $O/near_c1.log 2>&1; cat $O/near_c1.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log

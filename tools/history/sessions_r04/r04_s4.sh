#!/bin/bash
# round 4, session 4: one SDWA instruction per table address + integer LDS addressing + one near mask (hot path 490 -> 482 / 581 -> 560
# instructions per group): A/B against the round-3 kernel in BOTH orders (the first context of a process is not the second), then the suite
set -o pipefail
O=gpurun_out/r4s4; mkdir -p $O
L=chan_vese_amd/csrc
for rep in 1 2; do
timeout -k 10 300 python tools/ab_libs.py $L/variants/orig/libchanvese_hip.so $L/libchanvese_hip.so >> $O/ab_c1.log 2>&1
timeout -k 10 300 python tools/ab_libs.py $L/libchanvese_hip.so $L/variants/orig/libchanvese_hip.so >> $O/ab_c1.log 2>&1
C=3 timeout -k 10 300 python tools/ab_libs.py $L/variants/orig/libchanvese_hip.so $L/libchanvese_hip.so >> $O/ab_c3.log 2>&1
C=3 timeout -k 10 300 python tools/ab_libs.py $L/libchanvese_hip.so $L/variants/orig/libchanvese_hip.so >> $O/ab_c3.log 2>&1
done
cat $O/ab_c1.log $O/ab_c3.log
N=2048 OPTS=resident=1 timeout -k 10 300 python tools/ab_libs.py $L/libchanvese_hip.so $L/variants/orig/libchanvese_hip.so > $O/ab_2048_resident.log 2>&1; cat $O/ab_2048_resident.log
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log

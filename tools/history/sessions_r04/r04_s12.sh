#!/bin/bash
# round 4, session 12: the one-reciprocal near form (heaviside_centred_near1) in the 2-pixel kernel: near regime, far regime A/B, suite
set -o pipefail
O=gpurun_out/r4s12; mkdir -p $O
L=chan_vese_amd/csrc; V=$L/variants
RESIDENT=0 timeout -k 10 400 python tools/near_regime_probe.py > $O/near_c1.log 2>&1; cat $O/near_c1.log
C=3 SIZES=4096 timeout -k 10 300 python tools/near_regime_probe.py > $O/near_c3.log 2>&1; cat $O/near_c3.log
REPS=3 timeout -k 10 300 python tools/ab_libs.py $V/orig/libchanvese_hip.so $L/libchanvese_hip.so > $O/ab_c1.log 2>&1; cat $O/ab_c1.log
REPS=3 timeout -k 10 300 python tools/ab_libs.py $L/libchanvese_hip.so $V/orig/libchanvese_hip.so >> $O/ab_c1.log 2>&1; tail -2 $O/ab_c1.log
C=3 REPS=3 timeout -k 10 300 python tools/ab_libs.py $V/orig/libchanvese_hip.so $L/libchanvese_hip.so > $O/ab_c3.log 2>&1; cat $O/ab_c3.log
C=3 REPS=3 timeout -k 10 300 python tools/ab_libs.py $L/libchanvese_hip.so $V/orig/libchanvese_hip.so >> $O/ab_c3.log 2>&1; tail -2 $O/ab_c3.log
timeout -k 10 300 python bench.py --config near --no-cpu-baseline > $O/bench_near.json 2> $O/bench_near.err; cut -c1-120 $O/bench_near.json
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -6 $O/pytest.log

#!/bin/bash
# round 4, session 3: the near-form march without spills (table forms behind the park): far regime A/B against the round-3 kernel,
# near regime per flow, then the GPU suite
set -o pipefail
O=gpurun_out/r4s3; mkdir -p $O
L=chan_vese_amd/csrc
timeout -k 10 300 python tools/ab_libs.py $L/variants/orig/libchanvese_hip.so $L/libchanvese_hip.so > $O/ab_c1.log 2>&1; cat $O/ab_c1.log
C=3 timeout -k 10 300 python tools/ab_libs.py $L/libchanvese_hip.so $L/variants/orig/libchanvese_hip.so > $O/ab_c3.log 2>&1; cat $O/ab_c3.log
RESIDENT=0 timeout -k 10 400 python tools/near_regime_probe.py > $O/near_c1.log 2>&1; cat $O/near_c1.log
SIZES=2048 timeout -k 10 400 python tools/near_regime_probe.py > $O/near_c1_resident.log 2>&1; cat $O/near_c1_resident.log
C=3 SIZES=4096 timeout -k 10 300 python tools/near_regime_probe.py > $O/near_c3.log 2>&1; cat $O/near_c3.log
timeout -k 10 300 python bench.py --config near --no-cpu-baseline > $O/bench_near.json 2> $O/bench_near.err; cut -c1-600 $O/bench_near.json; tail -3 $O/bench_near.err
timeout -k 10 1000 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log

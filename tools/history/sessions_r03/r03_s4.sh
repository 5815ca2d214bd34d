#!/bin/bash
# round 3, session 4: FP64 / f32 transcendental issue rates; A/B of the SGPR-addend and f32-seed builds; measured-balance proof of concept
set -o pipefail
O=gpurun_out/r3s4; mkdir -p $O
timeout -k 10 200 tools/dp_rate_probe > $O/dp_rate.txt 2>&1; grep -E "waves/SIMD=3|device" $O/dp_rate.txt
V=chan_vese_amd/csrc/variants; D=chan_vese_amd/csrc/libchanvese_hip.so
REPS=4 timeout -k 10 300 python tools/ab_libs.py $D $V/fma3s/libchanvese_hip.so $V/f32seed/libchanvese_hip.so $V/both/libchanvese_hip.so $D > $O/ab_libs.txt 2>&1; cat $O/ab_libs.txt
N=2048 REPS=4 timeout -k 10 300 python tools/ab_libs.py $D $V/fma3s/libchanvese_hip.so $V/f32seed/libchanvese_hip.so $V/both/libchanvese_hip.so > $O/ab_libs_2048.txt 2>&1; cat $O/ab_libs_2048.txt
N=4096 ROUNDS=3 K=4 timeout -k 10 300 python tools/balance_poc.py > $O/balance_poc.txt 2>&1; cat $O/balance_poc.txt
N=4096 ROUNDS=3 K=4 timeout -k 10 300 python tools/balance_poc.py wave_cskew=0 > $O/balance_poc_noskew.txt 2>&1; tail -6 $O/balance_poc_noskew.txt

#!/bin/bash
# round 3, session 31: master with two polls in flight vs one, one process
set -o pipefail
O=gpurun_out/r3s31; mkdir -p $O
V=chan_vese_amd/csrc/variants
N=2048 REPS=5 STEPS=1024 timeout -k 10 300 python tools/ab_libs.py chan_vese_amd/csrc/libchanvese_hip.so $V/mpipe/libchanvese_hip.so > $O/ab.log 2>&1; cat $O/ab.log
N=1024 REPS=5 STEPS=1024 timeout -k 10 300 python tools/ab_libs.py chan_vese_amd/csrc/libchanvese_hip.so $V/mpipe/libchanvese_hip.so > $O/ab1024.log 2>&1; cat $O/ab1024.log

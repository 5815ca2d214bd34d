#!/bin/bash
# round 3, session 32: fewer polls: sleep between the master's / the workgroups' polls, one process
set -o pipefail
O=gpurun_out/r3s32; mkdir -p $O
V=chan_vese_amd/csrc/variants
N=2048 REPS=5 STEPS=1024 timeout -k 10 400 python tools/ab_libs.py chan_vese_amd/csrc/libchanvese_hip.so $V/m8/libchanvese_hip.so $V/m24/libchanvese_hip.so $V/g12/libchanvese_hip.so $V/g32/libchanvese_hip.so $V/m8g12/libchanvese_hip.so > $O/ab.log 2>&1; cat $O/ab.log

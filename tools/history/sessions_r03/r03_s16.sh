#!/bin/bash
# round 3, session 16: same-box A/B of the resident kernel: previous commit vs sums-in-arrival-lines + barrier-free master, +- pipelined polls
set -o pipefail
O=gpurun_out/r3s16; mkdir -p $O
V=chan_vese_amd/csrc/variants
N=2048 REPS=5 STEPS=1024 timeout -k 10 300 python tools/ab_libs.py $V/prev/libchanvese_hip.so chan_vese_amd/csrc/libchanvese_hip.so $V/pipe2/libchanvese_hip.so $V/pipe4/libchanvese_hip.so > $O/ab.log 2>&1; cat $O/ab.log

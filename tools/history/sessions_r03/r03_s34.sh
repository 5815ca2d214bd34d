#!/bin/bash
# round 3, session 34: resident Perona-Malik with two march instances (full tile: no ring factor on inner rows) vs the committed build
set -o pipefail
O=gpurun_out/r3s34; mkdir -p $O
V=chan_vese_amd/csrc/variants
OPTS=pm_kernel=4 N=2048 timeout -k 10 300 python tools/pm_ab_libs.py $V/pmhead/libchanvese_hip.so chan_vese_amd/csrc/libchanvese_hip.so > $O/ab2048.log 2>&1; cat $O/ab2048.log
OPTS=pm_kernel=4 N=1024 timeout -k 10 300 python tools/pm_ab_libs.py $V/pmhead/libchanvese_hip.so chan_vese_amd/csrc/libchanvese_hip.so > $O/ab1024.log 2>&1; cat $O/ab1024.log
OPTS=pm_kernel=4 N=512 timeout -k 10 300 python tools/pm_ab_libs.py $V/pmhead/libchanvese_hip.so chan_vese_amd/csrc/libchanvese_hip.so > $O/ab512.log 2>&1; cat $O/ab512.log

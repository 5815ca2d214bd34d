#!/bin/bash
# round 3, session 26: scheduling variants of the resident Perona-Malik kernel in one process
set -o pipefail
O=gpurun_out/r3s26; mkdir -p $O
V=chan_vese_amd/csrc/variants
OPTS=pm_kernel=4 timeout -k 10 300 python tools/pm_ab_libs.py chan_vese_amd/csrc/libchanvese_hip.so $V/ilp/libchanvese_hip.so $V/sb4/libchanvese_hip.so $V/sbnone/libchanvese_hip.so $V/sb1/libchanvese_hip.so $V/ilpsb4/libchanvese_hip.so > $O/ab.log 2>&1; cat $O/ab.log

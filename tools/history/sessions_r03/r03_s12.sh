#!/bin/bash
# round 3, session 12: 3-channel 2-pixel kernel with the samples read one row ahead (A/B of builds); smoke (both flows)
set -o pipefail
O=gpurun_out/r3s12; mkdir -p $O
V=chan_vese_amd/csrc/variants; D=chan_vese_amd/csrc/libchanvese_hip.so
C=3 N=4096 REPS=3 OPTS=kernel=3 timeout -k 10 300 python tools/ab_libs.py $D $V/c3ahead/libchanvese_hip.so $D > $O/ab_c3ahead.txt 2>&1; cat $O/ab_c3ahead.txt
C=3 N=4096 REPS=3 OPTS=kernel=2 timeout -k 10 300 python tools/ab_libs.py $D > $O/ab_c3_1px.txt 2>&1; cat $O/ab_c3_1px.txt
timeout -k 10 120 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log

#!/bin/bash
# round 4, session 60: the poll cadences of the resident CSV kernel once more, after everything else changed (builds with other constants, one process)
set -o pipefail
O=gpurun_out/r4s60; mkdir -p $O
L=tools/experiments/_libs
for n in 2048 1024; do
N=$n REPS=4 OPTS=resident=1 timeout -k 10 500 python tools/ab_libs.py $L/lib_m1_r3.so $L/lib_m0_r3.so $L/lib_m2_r3.so $L/lib_m1_r1.so $L/lib_m1_r6.so $L/lib_m1_r0.so > $O/ab_libs_$n.log 2>&1; cat $O/ab_libs_$n.log
done

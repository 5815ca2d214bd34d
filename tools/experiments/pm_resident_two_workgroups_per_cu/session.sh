#!/bin/bash
# round 4, session 44: resident Perona-Malik with TWO 4-wave workgroups per CU (the two waves of a SIMD belong to different tiles)
set -o pipefail
O=gpurun_out/r4s44; mkdir -p $O
N=256 STEPS=40 REPS=1 timeout -k 10 120 python tools/pm_ab_opts.py "pm_kernel=4" "pm_kernel=4,pm_res_waves=4" > $O/small.log 2>&1; cat $O/small.log
grep -q "differs\|Error\|error" $O/small.log && exit 1
for n in 2048 1024 512; do
N=$n REPS=4 timeout -k 10 300 python tools/pm_ab_opts.py "pm_kernel=4" "pm_kernel=4,pm_res_waves=4" > $O/pm_ab_$n.log 2>&1; cat $O/pm_ab_$n.log
done
H=1200 W=1920 REPS=3 timeout -k 10 300 python tools/pm_ab_opts.py "pm_kernel=4" "pm_kernel=4,pm_res_waves=4" > $O/pm_ab_1200x1920.log 2>&1; cat $O/pm_ab_1200x1920.log
MATH=1 N=2048 REPS=2 timeout -k 10 300 python tools/pm_ab_opts.py "pm_kernel=4" "pm_kernel=4,pm_res_waves=4" > $O/pm_ab_2048_strict.log 2>&1; cat $O/pm_ab_2048_strict.log

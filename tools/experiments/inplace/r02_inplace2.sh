#!/bin/bash
mkdir -p gpurun_out/inplace
REPS=4 timeout -k 10 300 python tools/ab_probe.py "inplace=0" "inplace=1" "inplace=1,wave_cskew=350" "inplace=1,wave_cskew=650" > gpurun_out/inplace/ab4096.log 2>&1; echo "rc=$?"; cat gpurun_out/inplace/ab4096.log
N=2048 REPS=3 timeout -k 10 300 python tools/ab_probe.py "inplace=0" "inplace=1" > gpurun_out/inplace/ab2048.log 2>&1; cat gpurun_out/inplace/ab2048.log
H=3000 W=4000 REPS=3 timeout -k 10 300 python tools/ab_probe.py "inplace=0" "inplace=1" > gpurun_out/inplace/ab3000.log 2>&1; cat gpurun_out/inplace/ab3000.log
N=5120 REPS=3 timeout -k 10 300 python tools/ab_probe.py "inplace=0" "inplace=1" > gpurun_out/inplace/ab5120.log 2>&1; cat gpurun_out/inplace/ab5120.log

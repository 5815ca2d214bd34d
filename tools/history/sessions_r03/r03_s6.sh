#!/bin/bash
# round 3, session 6: exact strip counts on cache-resident planes (2-pixel kernel): how many workgroups should a small plane be cut into?
set -o pipefail
O=gpurun_out/r3s6; mkdir -p $O
N=2048 REPS=3 timeout -k 10 400 python tools/ab_probe.py "strips=0" "strips=56" "strips=84" "strips=100" "strips=104" "strips=108" "strips=112" "strips=114" "strips=128" "strips=140" "strips=168" "strips=56,wave_cskew=0" "strips=112,wave_cskew=0" "strips=168,wave_cskew=0" > $O/strips2048.txt 2>&1; cat $O/strips2048.txt
N=1024 REPS=3 timeout -k 10 400 python tools/ab_probe.py "kernel=3,strips=0" "kernel=3,strips=32" "kernel=3,strips=56" "kernel=3,strips=64" "kernel=3,strips=96" "kernel=3,strips=102" "kernel=2" > $O/strips1024.txt 2>&1; cat $O/strips1024.txt
N=3072 REPS=3 timeout -k 10 400 python tools/ab_probe.py "strips=0" "strips=112" "strips=116" "strips=118" "strips=120" "strips=124" > $O/strips3072.txt 2>&1; cat $O/strips3072.txt

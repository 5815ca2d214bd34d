#!/bin/bash
# round 3, session 7: what would 4 waves/SIMD buy? (existing 128-VGPR flavours with spills, per-row branch)
set -o pipefail
O=gpurun_out/r3s7; mkdir -p $O
N=4096 REPS=3 timeout -k 10 300 python tools/ab_probe.py "kernel=3" "kernel=3,wave_occupancy=4" "kernel=2,wave_occupancy=5" "kernel=2,wave_occupancy=4" > $O/occ4096.txt 2>&1; cat $O/occ4096.txt
N=2048 REPS=3 timeout -k 10 300 python tools/ab_probe.py "kernel=3" "kernel=3,wave_occupancy=4" "kernel=2,wave_occupancy=5" "kernel=2,wave_occupancy=4" > $O/occ2048.txt 2>&1; cat $O/occ2048.txt

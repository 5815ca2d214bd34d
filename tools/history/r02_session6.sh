#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s6
# process-to-process variance at identical settings, and the effect of the skew between the two level-set buffers
for rep in 1 2 3; do
for sk in 0 256 4096 65536 1048576; do
  CVH_U_SKEW=$sk REPS=2 python tools/ab_probe.py wave_cskew=0 2>&1 | sed "s/^/skew $sk rep $rep: /" | tee -a gpurun_out/s6/uskew.log
done
done
python tools/ab_probe.py "wave_cskew=0,wave_prio=0" "wave_cskew=0,wave_prio=1" "wave_cskew=0,wave_prio=2" "wave_cskew=0,wave_prio=3" \
  "wave_cskew=100,wave_prio=0" "wave_cskew=100,wave_prio=1" "wave_cskew=100,wave_prio=2" "wave_cskew=100,wave_prio=3" \
  "wave_cskew=160,wave_prio=0" "wave_cskew=160,wave_prio=1" "wave_cskew=160,wave_prio=2" "wave_cskew=160,wave_prio=3" \
  "wave_cskew=220,wave_prio=0" "wave_cskew=220,wave_prio=2" "wave_cskew=300,wave_prio=2" > gpurun_out/s6/ab_prio.log 2>&1; echo "abp rc=$?"; cat gpurun_out/s6/ab_prio.log
N=2048 python tools/ab_probe.py strip_rows=0 strip_rows=16 strip_rows=20 strip_rows=24 strip_rows=32 strip_rows=44 "strip_rows=32,wave_prio=0" > gpurun_out/s6/ab_2048.log 2>&1; echo "ab2048 rc=$?"; cat gpurun_out/s6/ab_2048.log

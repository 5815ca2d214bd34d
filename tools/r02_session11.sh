#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s11
N=2048 python tools/ab_probe.py "strip_rows=0" "wave_sync=0" "wave_prio=0" "wave_occupancy=4" "wave_occupancy=4,strip_rows=16" "kernel=2" "kernel=2,strip_rows=16" "wave_cskew=0" "chain=0" > gpurun_out/s11/ab_2048.log 2>&1; cat gpurun_out/s11/ab_2048.log
N=1024 python tools/ab_probe.py "strip_rows=0" "wave_sync=0" "wave_occupancy=4" "kernel=2" "wave_cskew=0" > gpurun_out/s11/ab_1024.log 2>&1; cat gpurun_out/s11/ab_1024.log
N=512 python tools/ab_probe.py "strip_rows=0" "wave_sync=0" "wave_occupancy=4" "kernel=2" "wave_cskew=0" > gpurun_out/s11/ab_512.log 2>&1; cat gpurun_out/s11/ab_512.log

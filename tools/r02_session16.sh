#!/bin/bash
mkdir -p gpurun_out/s16
python tools/dbg_c1occ.py 4096 > gpurun_out/s16/dbg_c1.log 2>&1; cat gpurun_out/s16/dbg_c1.log
python tools/dbg_c3b.py > gpurun_out/s16/dbg_c3.log 2>&1; grep "4096\|2048" gpurun_out/s16/dbg_c3.log
python tools/ab_probe.py wave_cskew=500 > gpurun_out/s16/ab_c2.log 2>&1; cat gpurun_out/s16/ab_c2.log
C=3 python tools/ab_probe.py kernel=2 kernel=3 "kernel=3,wave_cskew=300" "kernel=3,wave_cskew=0" > gpurun_out/s16/ab_c3.log 2>&1; cat gpurun_out/s16/ab_c3.log

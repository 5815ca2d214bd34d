#!/bin/bash
# does the 20-step line suffer right after a CPU-heavy process (as pytest's oracle runs are)?
mkdir -p gpurun_out/al
: > gpurun_out/al/summary.txt
one() { python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-phases > gpurun_out/al/o.json 2>gpurun_out/al/e.log
        python -c "import json;d=json.load(open('gpurun_out/al/o.json'));print('$1: avg launch %.2f us, wall %.2f us' % (d['roofline']['avg_launch_us'], d['ms_per_step']*1e3))" >> gpurun_out/al/summary.txt; }
one "fresh"
for r in 1 2 3; do
  python -c "
import sys; sys.path.insert(0,'.')
import numpy as np, time
from oracle import cv_oracle as o
from chan_vese_amd import synth
img = synth.disk(4096); u = o.checkerboard(4096, 4096)
t=time.time()
while time.time()-t < 40: u = o.csv_step([img], u, o.make_params(tol=0))[0]
" > gpurun_out/al/cpu.log 2>&1
  one "after 40 s of the CPU oracle ($r)"
  one "next process ($r)"
done
cat gpurun_out/al/summary.txt

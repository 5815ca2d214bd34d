#!/bin/bash
# distribution of the driver's command over ten fresh processes on one box
mkdir -p gpurun_out/dist
: > gpurun_out/dist/summary.txt
for r in 1 2 3 4 5 6 7 8 9 10; do
  python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/dist/run_$r.json 2>gpurun_out/dist/e.log
  python -c "import json;d=json.load(open('gpurun_out/dist/run_$r.json'));print('run $r: avg launch %.2f us, ms_per_step %.4f, value %.0f, frac %.4f' % (d['roofline']['avg_launch_us'], d['ms_per_step'], d['value'], d['roofline']['frac']))" >> gpurun_out/dist/summary.txt
done
cat gpurun_out/dist/summary.txt

#!/bin/bash
mkdir -p gpurun_out/bd
for r in 1 2 3 4 5; do
  python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bd/o.json 2>gpurun_out/bd/e.log
  python -c "import json;d=json.load(open('gpurun_out/bd/o.json'));print('new order:', round(d['roofline']['avg_launch_us'],2), round(d['ms_per_step']*1e3,2))"
  python tools/_bench_old.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/bd/o.json 2>gpurun_out/bd/e.log
  python -c "import json;d=json.load(open('gpurun_out/bd/o.json'));print('old order:', round(d['roofline']['avg_launch_us'],2), round(d['ms_per_step']*1e3,2))"
done

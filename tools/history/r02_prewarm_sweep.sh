#!/bin/bash
mkdir -p gpurun_out/pw
for rep in 1 2; do for pw in 0 20 60 150 400 1000; do
  python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --no-phases --prewarm-ms $pw > gpurun_out/pw/pw_${pw}_$rep.json 2>/dev/null
  python - gpurun_out/pw/pw_${pw}_$rep.json $pw <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print('prewarm %s ms: ms/step %.4f launch_us %.2f frac %.3f'%(sys.argv[2],d['ms_per_step'],d['roofline']['avg_launch_us'],d['roofline']['frac']))
PY
done; done

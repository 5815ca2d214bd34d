#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s8
python -m pytest tests -m gpu -q -x > gpurun_out/s8/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/s8/pytest_all.log
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/s8/bench_driver.json 2>gpurun_out/s8/bench_driver.err; echo "driver rc=$?"
python bench.py > gpurun_out/s8/bench_default.json 2>gpurun_out/s8/bench_default.err; echo "default rc=$?"
python bench.py --config C3 > gpurun_out/s8/bench_C3.json 2>&1; echo "C3 rc=$?"
python bench.py --config C4 > gpurun_out/s8/bench_C4.json 2>&1; echo "C4 rc=$?"
python bench.py --config C5 --no-cpu-baseline > gpurun_out/s8/bench_C5.json 2>&1; echo "C5 rc=$?"
python bench.py --config C5-image --no-cpu-baseline > gpurun_out/s8/bench_C5i.json 2>&1; echo "C5i rc=$?"
CHANVESE_DIST_BACKEND=gloo python bench.py --gpus 2 --steps 50 --warmup 10 > gpurun_out/s8/bench_gloo2.json 2>gpurun_out/s8/bench_gloo2.err; echo "gloo2 rc=$?"
N=4096 python tools/pm_ab.py pm_kernel=3 "pm_kernel=3,pm_strip_rows=80" "pm_kernel=3,pm_strip_rows=104" "pm_kernel=3,pm_strip_rows=128" "pm_kernel=3,pm_strip_rows=200" > gpurun_out/s8/pm_ab4096.log 2>&1; cat gpurun_out/s8/pm_ab4096.log
N=1024 python tools/pm_ab.py pm_kernel=1 pm_kernel=3 "pm_kernel=3,pm_strip_rows=16" "pm_kernel=3,pm_strip_rows=24" "pm_kernel=3,pm_strip_rows=32" > gpurun_out/s8/pm_ab1024.log 2>&1; cat gpurun_out/s8/pm_ab1024.log
N=512 python tools/pm_ab.py pm_kernel=1 pm_kernel=3 "pm_kernel=3,pm_strip_rows=16" "pm_kernel=3,pm_strip_rows=24" > gpurun_out/s8/pm_ab512.log 2>&1; cat gpurun_out/s8/pm_ab512.log
N=1024 python tools/ab_probe.py strip_rows=0 strip_rows=8 strip_rows=12 strip_rows=16 strip_rows=20 > gpurun_out/s8/ab_1024.log 2>&1; cat gpurun_out/s8/ab_1024.log
for f in driver default C3 C4 C5 C5i gloo2; do python - gpurun_out/s8/bench_$f.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[1].split('/')[-1], 'value %.0f ms/step %.4f launch_us %.2f frac %.3f'%(d['value'],d['ms_per_step'],d['roofline']['avg_launch_us'],d['roofline']['frac']), d.get('phases'), (d.get('pm') or {}).get('us_per_step'), ((d.get('pm') or {}).get('roofline') or {}).get('frac'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done

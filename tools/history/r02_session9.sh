#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s9
python -m pytest tests -m gpu -q -x > gpurun_out/s9/pytest_all.log 2>&1; echo "pytest rc=$?"; tail -4 gpurun_out/s9/pytest_all.log
C=3 python tools/ab_probe.py "chain=1,wave_cskew=500" "chain=0,wave_cskew=500" "chain=1,wave_cskew=0" "chain=1,wave_cskew=300" "chain=1,wave_cskew=800" "chain=1,wave_cls=0" > gpurun_out/s9/ab_c3.log 2>&1; echo "abc3 rc=$?"; cat gpurun_out/s9/ab_c3.log
python bench.py --config C3 --no-cpu-baseline > gpurun_out/s9/bench_C3.json 2>&1; echo "C3 rc=$?"
python bench.py --config C4 --no-cpu-baseline > gpurun_out/s9/bench_C4.json 2>&1; echo "C4 rc=$?"
N=2048 python tools/ab_probe.py strip_rows=0 strip_rows=20 strip_rows=16 > gpurun_out/s9/ab_2048.log 2>&1; cat gpurun_out/s9/ab_2048.log
N=1000 python tools/ab_probe.py "kernel=2,wave_cskew=500" "kernel=2,wave_cskew=0" "kernel=2,wave_cls=0" "kernel=2,chain=0" > gpurun_out/s9/ab_k2_1000.log 2>&1; cat gpurun_out/s9/ab_k2_1000.log
python tools/ab_probe.py "kernel=2,wave_cskew=500" "kernel=2,wave_cskew=0" "kernel=2,wave_cls=0" "kernel=2,chain=0" "kernel=3" > gpurun_out/s9/ab_k2_4096.log 2>&1; cat gpurun_out/s9/ab_k2_4096.log
for f in C3 C4; do python - gpurun_out/s9/bench_$f.json <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[1].split('/')[-1], 'value %.0f ms/step %.4f launch_us %.2f frac %.3f'%(d['value'],d['ms_per_step'],d['roofline']['avg_launch_us'],d['roofline']['frac']), d.get('phases'), (d.get('pm') or {}).get('us_per_step'), ((d.get('pm') or {}).get('roofline') or {}).get('frac'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done

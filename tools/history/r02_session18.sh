#!/bin/bash
mkdir -p gpurun_out/s18
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "stop or continu or config2 or unlimited or interleav or config1" > gpurun_out/s18/pytest_sub.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s18/pytest_sub.log
for i in 1 2 3; do python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/s18/driver_$i.json 2>&1; python - gpurun_out/s18/driver_$i.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print('driver cmd: ms/step %.4f launch_us %.2f frac %.3f'%(d['ms_per_step'],d['roofline']['avg_launch_us'],d['roofline']['frac']), d.get('phases'))
PY
done
python bench.py --no-cpu-baseline > gpurun_out/s18/default.json 2>&1; python - gpurun_out/s18/default.json <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1]); print('default: ms/step %.4f launch_us %.2f frac %.3f'%(d['ms_per_step'],d['roofline']['avg_launch_us'],d['roofline']['frac']), d.get('phases'))
PY

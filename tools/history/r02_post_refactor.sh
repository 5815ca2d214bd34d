#!/bin/bash
mkdir -p gpurun_out/post
python -m pytest tests/test_gpu_parity.py tests/test_gpu_fullsize.py -m gpu -q -x -k "two_pixel or kernel_variants or config2 or flavours or three_channel" > gpurun_out/post/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/post/pytest.log
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/post/bench.json 2>/dev/null; python - <<'PY'
import json
d=json.loads([l for l in open('gpurun_out/post/bench.json') if l.startswith('{')][-1]); print('driver cmd: ms/step %.4f launch_us %.2f frac %.3f'%(d['ms_per_step'],d['roofline']['avg_launch_us'],d['roofline']['frac']), d['config']['device_prewarm'])
PY

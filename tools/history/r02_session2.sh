#!/bin/bash
set -o pipefail
mkdir -p gpurun_out/s2
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "two_pixel or kernel_variants or small_shapes or stop_rule or config1" > gpurun_out/s2/pytest_sub.log 2>&1; echo "pytest rc=$?"; tail -3 gpurun_out/s2/pytest_sub.log
python tools/fullsize_diag.py > gpurun_out/s2/diag.log 2>&1; echo "diag rc=$?"
B="python bench.py --no-cpu-baseline"
for a in 0 40 70 100 130; do
  $B --opt wave_cskew=$a > gpurun_out/s2/skew_$a.json 2> gpurun_out/s2/skew_$a.err; echo "skew $a rc=$?"
done
$B --opt wave_cls=0 > gpurun_out/s2/cls0.json 2>&1; echo "cls0 rc=$?"
CHANVESE_HIP_LIB=$PWD/chan_vese_amd/csrc/variants/far4/libchanvese_hip.so $B --opt far_terms=4 > gpurun_out/s2/far4.json 2>&1; echo "far4 rc=$?"
CHANVESE_HIP_LIB=$PWD/chan_vese_amd/csrc/variants/far4/libchanvese_hip.so $B --opt far_terms=4 --opt wave_cskew=70 > gpurun_out/s2/far4_skew70.json 2>&1; echo "far4s rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/s2/driver.json 2>&1; echo "driver rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --opt wave_cskew=70 > gpurun_out/s2/driver_skew70.json 2>&1; echo "driver70 rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline --prewarm-ms 0 > gpurun_out/s2/driver_nopw.json 2>&1; echo "driver nopw rc=$?"
KERNEL=3 ITERS=40 SAVE=gpurun_out/s2/timeline_skew70.npz python tools/wave_timeline.py wave_cskew=70 > gpurun_out/s2/timeline_skew70.log 2>&1; echo "tl rc=$?"
KERNEL=3 ITERS=40 SAVE=gpurun_out/s2/timeline_skew0.npz python tools/wave_timeline.py > gpurun_out/s2/timeline_skew0.log 2>&1; echo "tl0 rc=$?"
for f in gpurun_out/s2/*.json; do python - "$f" <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
    print(sys.argv[1].split('/')[-1], 'ms/step %.4f launch_us %.2f frac %.3f'%(d['ms_per_step'],d['roofline']['avg_launch_us'],d['roofline']['frac']), d.get('phases'))
except Exception as e: print(sys.argv[1], 'ERR', e)
PY
done

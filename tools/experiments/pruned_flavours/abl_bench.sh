#!/bin/bash
# diagnostic: time the wave kernel with the arithmetic or the global memory traffic compiled out
for v in COMPUTE MEMORY; do
  cp chan_vese_amd/csrc/libchanvese_hip.so /tmp/lib_keep.so
  cp chan_vese_amd/csrc/libchanvese_abl_$v.so chan_vese_amd/csrc/libchanvese_hip.so
  python bench.py --steps 200 --warmup 10 --no-cpu-baseline --opt wave_depth=8 2>&1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate $v', round(d['roofline']['avg_launch_us'],1))"
  cp /tmp/lib_keep.so chan_vese_amd/csrc/libchanvese_hip.so
done

#!/bin/bash
# diagnostic: bench build variants chan_vese_amd/csrc/libchanvese_var_<name>.so of the library
cp chan_vese_amd/csrc/libchanvese_hip.so /tmp/lib_keep.so
for f in chan_vese_amd/csrc/libchanvese_var_*.so; do
  n=$(basename $f .so); n=${n#libchanvese_var_}
  cp $f chan_vese_amd/csrc/libchanvese_hip.so
  for rep in 1 2; do
  timeout -k 10 120 python bench.py --steps 300 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant $n', round(d['roofline']['avg_launch_us'],2))"
  done
done
cp /tmp/lib_keep.so chan_vese_amd/csrc/libchanvese_hip.so
for rep in 1 2; do
timeout -k 10 120 python bench.py --steps 300 --warmup 20 --no-cpu-baseline "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('default', round(d['roofline']['avg_launch_us'],2))"
done

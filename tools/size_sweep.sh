#!/bin/bash
# diagnostic: tools/size_sweep.sh SIZE "k=v,..." ...  -> bench.py --size SIZE under each option set
size=$1; shift
for o in "$@"; do
  args=""; for kv in ${o//,/ }; do args="$args --opt $kv"; done
  [ "$o" = "-" ] && args=""
  timeout -k 10 120 python bench.py --size $size --steps 300 --warmup 20 --no-cpu-baseline $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$size $o', round(d['roofline']['avg_launch_us'],2), round(d['roofline']['frac'],3))"
done

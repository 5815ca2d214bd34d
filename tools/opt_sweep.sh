#!/bin/bash
# diagnostic: bench.py under a list of "--opt k=v[,k=v]" settings:  tools/opt_sweep.sh "wave_skew=0" "wave_skew=80" ...
for o in "$@"; do
  args=""; for kv in ${o//,/ }; do args="$args --opt $kv"; done
  [ "$o" = "-" ] && args=""
  for rep in 1 2; do
    timeout -k 10 120 python bench.py --steps 300 --warmup 20 --no-cpu-baseline $args 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$o', round(d['roofline']['avg_launch_us'],2), round(d['roofline']['frac'],3))"
  done
done

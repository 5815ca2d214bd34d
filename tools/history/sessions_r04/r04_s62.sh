#!/bin/bash
# round 4, session 62: batch of eight planes, sizes between 1024^2 and 2048^2
set -o pipefail
O=gpurun_out/r4s62; mkdir -p $O
for n in 1536 1792 1280; do
N=$n REPS=2 timeout -k 10 300 python tools/batch_probe.py > $O/batch_$n.log 2>&1; cat $O/batch_$n.log
done

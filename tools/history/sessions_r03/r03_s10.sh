#!/bin/bash
# round 3, session 10: resident vs per-launch flow over plane sizes (auto threshold)
set -o pipefail
O=gpurun_out/r3s10; mkdir -p $O
for hw in "128 128" "256 256" "384 512" "768 768" "1024 2048" "1536 1536" "2048 1024" "1200 1920" "2048 2048"; do set -- $hw; H=$1 W=$2 N=$2 REPS=3 STEPS=200 timeout -k 10 200 python tools/ab_probe.py "resident=0" "resident=1" >> $O/sizes.txt 2>&1; done
cat $O/sizes.txt

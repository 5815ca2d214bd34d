#!/bin/bash
mkdir -p gpurun_out/persist
for n in 4096; do N=$n timeout -k 10 120 python tools/persist_timeline.py > gpurun_out/persist/timeline_$n.log 2>&1; echo "rc=$?"; cat gpurun_out/persist/timeline_$n.log; done

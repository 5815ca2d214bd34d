#!/bin/bash
# round 3, session 17: what do the pixels near the contour cost (streaming 4096^2, resident 2048^2)?
set -o pipefail
O=gpurun_out/r3s17; mkdir -p $O
timeout -k 10 300 python tools/near_cost_probe.py > $O/near_4096.log 2>&1; cat $O/near_4096.log
N=2048 RESIDENT=1 STEPS=1024 timeout -k 10 300 python tools/near_cost_probe.py > $O/near_2048r.log 2>&1; cat $O/near_2048r.log

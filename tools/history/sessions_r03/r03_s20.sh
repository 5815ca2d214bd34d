#!/bin/bash
# round 3, session 20: where a step of the resident Perona-Malik kernel goes
set -o pipefail
O=gpurun_out/r3s20; mkdir -p $O
N=2048 timeout -k 10 200 python tools/pm_resident_timeline.py > $O/tl_2048.log 2>&1; cat $O/tl_2048.log
N=1024 timeout -k 10 200 python tools/pm_resident_timeline.py > $O/tl_1024.log 2>&1; cat $O/tl_1024.log

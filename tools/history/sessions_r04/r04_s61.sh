#!/bin/bash
# round 4, session 61: a batch of eight planes that fit the chip, once more with the faster resident kernel: per-launch interleaved vs resident launches one after the other
set -o pipefail
O=gpurun_out/r4s61; mkdir -p $O
N=2048 timeout -k 10 400 python tools/batch_probe.py > $O/batch_2048.log 2>&1; cat $O/batch_2048.log
N=1024 timeout -k 10 400 python tools/batch_probe.py > $O/batch_1024.log 2>&1; cat $O/batch_1024.log

#!/bin/bash
# round 3, session 5: measured balance, second attempt: strips scaled by the end time of the CU they ran on
set -o pipefail
O=gpurun_out/r3s5; mkdir -p $O
N=4096 ROUNDS=3 K=4 timeout -k 10 300 python tools/balance_poc.py > $O/balance_cu.txt 2>&1; cat $O/balance_cu.txt
N=4096 ROUNDS=5 K=6 DAMP=0.5 timeout -k 10 300 python tools/balance_poc.py > $O/balance_cu_slow.txt 2>&1; tail -9 $O/balance_cu_slow.txt

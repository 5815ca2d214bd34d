#!/bin/bash
# round 3, session 3: persistence of the wave-end imbalance; full GPU suite timing with 16 oracle threads
set -o pipefail
O=gpurun_out/r3s3; mkdir -p $O
N=4096 K=8 SAVE=$O/imb4096.npz timeout -k 10 200 python tools/imbalance_probe.py > $O/imb4096.txt 2>&1; cat $O/imb4096.txt
N=4096 K=8 timeout -k 10 200 python tools/imbalance_probe.py wave_cskew=0 > $O/imb4096_noskew.txt 2>&1; tail -9 $O/imb4096_noskew.txt
timeout -k 10 1000 python -m pytest tests -m gpu -x -q --durations=15 > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
tail -22 $O/pytest.log

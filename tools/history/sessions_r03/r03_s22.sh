#!/bin/bash
set -o pipefail
O=gpurun_out/r3s22; mkdir -p $O
for args in "37 130 2" "32 128 2" "16 128 2" "48 128 3" "37 256 2" "37 64 2"; do timeout -k 10 100 python tools/history/dbg_pm_resident.py $args 2>&1 | tail -3; done > $O/dbg.log 2>&1; cat $O/dbg.log

#!/bin/bash
# round 3: the evidence for profiles/r03_C2 .. r03_C5 (bench line with cpu_baseline + check, kernel trace stats of the same command, PMC passes)
set -o pipefail
for cfg in C2 C3 C4 C5; do
  args="--config $cfg"; [ $cfg = C2 ] && args=""
  bash tools/profile_round.sh r03_$cfg "$args" > gpurun_out/r03_${cfg}_summary.txt 2>&1
  tail -12 gpurun_out/r03_${cfg}_summary.txt
done
# (C4 re-collected after resident mode became the automatic flow for planes that fit: bash tools/profile_round.sh r03_C4 "--config C4")

#!/bin/bash
# does the 32 KiB row pitch of a 4096-wide level set cost anything?  Same height, slightly wider images.
mkdir -p gpurun_out/pitch
for w in 4096 4112 4128 4160 4224 4352 4096 3968 4032; do
  H=4096 W=$w REPS=3 python tools/ab_probe.py "chain=1" >> gpurun_out/pitch/pitch.log 2>&1
done
cat gpurun_out/pitch/pitch.log

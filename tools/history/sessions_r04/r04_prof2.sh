#!/bin/bash
# round 4: the round's evidence, part 2 -- profiles/r04_C5, r04_near, r04_C2f32, r04_C3f32
bash tools/profile_round.sh r04_C5 "--config C5" > gpurun_out/r04_C5.summary.txt 2>&1; tail -8 gpurun_out/r04_C5.summary.txt
bash tools/profile_round.sh r04_near "--config near" > gpurun_out/r04_near.summary.txt 2>&1; tail -8 gpurun_out/r04_near.summary.txt
bash tools/profile_round.sh r04_C2f32 "--config C2-f32" > gpurun_out/r04_C2f32.summary.txt 2>&1; tail -8 gpurun_out/r04_C2f32.summary.txt
bash tools/profile_round.sh r04_C3f32 "--config C3-f32" > gpurun_out/r04_C3f32.summary.txt 2>&1; tail -8 gpurun_out/r04_C3f32.summary.txt

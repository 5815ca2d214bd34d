#!/bin/bash
# round 4: the round's evidence, part 1 -- profiles/r04_C2, r04_C3, r04_C4 (bench line + kernel trace of the same command + PMC passes)
bash tools/profile_round.sh r04_C2 "" > gpurun_out/r04_C2.summary.txt 2>&1; tail -12 gpurun_out/r04_C2.summary.txt
( cd $GRAFT_REPO_ROOT && timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 --prewarm-ms 0 > gpurun_out/r04_C2/bench_driver_cmd_no_prewarm.json 2> /dev/null )
bash tools/profile_round.sh r04_C3 "--config C3" > gpurun_out/r04_C3.summary.txt 2>&1; tail -8 gpurun_out/r04_C3.summary.txt
bash tools/profile_round.sh r04_C4 "--config C4" > gpurun_out/r04_C4.summary.txt 2>&1; tail -12 gpurun_out/r04_C4.summary.txt

#!/bin/bash
set -o pipefail
bash tools/profile_round.sh r02_C2 "" > gpurun_out/prof_r02_C2.log 2>&1; echo "prof C2 rc=$?"; tail -6 gpurun_out/prof_r02_C2.log
bash tools/profile_round.sh r02_C3 "--config C3" > gpurun_out/prof_r02_C3.log 2>&1; echo "prof C3 rc=$?"; tail -6 gpurun_out/prof_r02_C3.log
bash tools/profile_round.sh r02_C4 "--config C4" > gpurun_out/prof_r02_C4.log 2>&1; echo "prof C4 rc=$?"; tail -8 gpurun_out/prof_r02_C4.log
cd $GRAFT_REPO_ROOT
KERNEL=3 ITERS=40 SAVE=gpurun_out/r02_C2/timeline.npz python tools/wave_timeline.py > gpurun_out/r02_C2/wave_timeline_4096.txt 2>&1; echo "tl rc=$?"
python bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/r02_C2/bench_driver_cmd.json 2>/dev/null; echo "driver rc=$?"
python bench.py > gpurun_out/r02_C2/bench_default_500steps.json 2>/dev/null; echo "default rc=$?"
tail -c 600 gpurun_out/r02_C2/bench_driver_cmd.json

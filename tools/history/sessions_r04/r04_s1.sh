#!/bin/bash
# round 4, session 1: (a) the rocprofv3 + cooperative-launch exit() abort, isolated from the library (tools/coop_exit_probe.hip);
# (b) the all-near regime (dt = 0.001) against the default; (c) the round's starting bench lines
set -o pipefail
O=gpurun_out/r4s1; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for m in plain coop coop_null coop_leak; do
  $R/tools/coop_exit_probe $m > $R/$O/probe_bare_$m.log 2>&1; echo "bare $m rc=$?" | tee -a $R/$O/probe_rc.txt
  rm -rf /tmp/pk_$m
  timeout -k 10 120 rocprofv3 --kernel-trace -d /tmp/pk_$m -o p --output-format csv -- $R/tools/coop_exit_probe $m > $R/$O/probe_rocprof_$m.log 2>&1; echo "rocprofv3 $m rc=$?" | tee -a $R/$O/probe_rc.txt
done
cd $R
timeout -k 10 400 python tools/near_regime_probe.py > $O/near_c1.log 2>&1; cat $O/near_c1.log
C=3 SIZES=4096 timeout -k 10 300 python tools/near_regime_probe.py > $O/near_c3.log 2>&1; cat $O/near_c3.log
RESIDENT=0 SIZES=2048 timeout -k 10 300 python tools/near_regime_probe.py > $O/near_2048_perlaunch.log 2>&1; cat $O/near_2048_perlaunch.log
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/bench_driver.json 2> $O/bench_driver.err; cat $O/bench_driver.json | cut -c1-400
timeout -k 10 300 python bench.py --config C3 --no-cpu-baseline > $O/bench_C3.json 2> $O/bench_C3.err; cut -c1-300 $O/bench_C3.json
timeout -k 10 300 python bench.py --config C4 --no-cpu-baseline > $O/bench_C4.json 2> $O/bench_C4.err; cut -c1-300 $O/bench_C4.json

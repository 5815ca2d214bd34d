#!/bin/bash
# round 4, session 2: (a) owner of the faulting frame of the rocprofv3 + cooperative-launch exit() abort (probe prints its mappings);
# (b) the per-strip near-form march: far regime must not move (A/B against the round-3 kernel in one process), near regime should;
# (c) the GPU suite incl. the README example cases
set -o pipefail
O=gpurun_out/r4s2; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
timeout -k 10 120 rocprofv3 --kernel-trace -d /tmp/pk -o p --output-format csv -- $R/tools/coop_exit_probe coop maps > $R/$O/probe_rocprof_coop_maps.log 2>&1; echo "rocprofv3 coop maps rc=$?" | tee $R/$O/probe_rc.txt
cd $R
L=chan_vese_amd/csrc
timeout -k 10 300 python tools/ab_libs.py $L/variants/orig/libchanvese_hip.so $L/libchanvese_hip.so > $O/ab_c1.log 2>&1; cat $O/ab_c1.log
C=3 timeout -k 10 300 python tools/ab_libs.py $L/variants/orig/libchanvese_hip.so $L/libchanvese_hip.so > $O/ab_c3.log 2>&1; cat $O/ab_c3.log
RESIDENT=0 timeout -k 10 400 python tools/near_regime_probe.py > $O/near_c1.log 2>&1; cat $O/near_c1.log
C=3 SIZES=4096 timeout -k 10 300 python tools/near_regime_probe.py > $O/near_c3.log 2>&1; cat $O/near_c3.log
timeout -k 10 900 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log; tail -15 $O/pytest.log

#!/bin/bash
# round 3, session 23: counters of the resident Perona-Malik kernel at 2048^2
set -o pipefail
O=$GRAFT_REPO_ROOT/gpurun_out/r3s23; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
export SIZES=2048 FLOWS=4 REPS=1 STEPS=400
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" "SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_MFMA_MOPS_F64 SQ_THREAD_CYCLES_VALU"; do
  i=$((i+1)); rm -rf /tmp/pm_$i
  ( cd $GRAFT_REPO_ROOT && timeout -k 10 200 rocprofv3 --pmc $grp -d /tmp/pm_$i -o p --output-format csv -- python3 tools/pm_flows.py > /tmp/pm_$i.log 2>&1 ) || { echo "pass $i failed"; tail -3 /tmp/pm_$i.log; }
  f=$(find /tmp/pm_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY' >> $O/counters.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    if "pm_resident" in r["Kernel_Name"]: acc[r["Counter_Name"]][r["Dispatch_Id"]].append(float(r["Counter_Value"]))
for c, d in acc.items():
    v = [sum(x) for x in d.values()]
    print(c, "launches", len(v), "per launch", sum(v) / len(v), "per step (400)", sum(v) / len(v) / 400)
PY
done
cat $O/counters.txt

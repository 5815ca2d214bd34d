#!/bin/bash
# diagnostic: SQ counter passes of the default bench (one group per rocprofv3 run; no trace domains with --pmc)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" \
           "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM SQ_INSTS_SMEM" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA" \
           "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64" \
           "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS" \
           "SQ_CYCLES SQ_BUSY_CU_CYCLES SQ_LEVEL_WAVES SQ_WAVES GRBM_GUI_ACTIVE"; do
  i=$((i+1))
  rm -rf /tmp/pmc_$i
  ( cd $R && timeout -k 10 300 rocprofv3 --pmc $grp -d /tmp/pmc_$i -o p --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > /tmp/pmc_$i.log 2>&1 ) || { echo "pass $i failed"; tail -5 /tmp/pmc_$i.log; }
  f=$(find /tmp/pmc_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    if "csv_wave" in r["Kernel_Name"]:
        k = r["Counter_Name"]; acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for k, (v, n) in acc.items():
    print("%-28s per-launch %.4g  (launches %d)" % (k, v / n, n))
PY
done

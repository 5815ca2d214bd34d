#!/bin/bash
# diagnostic: HBM traffic counters of the default bench, one counter per pass (MI355X_MICROARCH.md: FETCH_SIZE
# reads 1/2 of the streamed bytes on gfx950; unit KiB)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "TCC_EA_RDREQ_sum TCC_EA_RDREQ_32B_sum" "TCC_EA_WRREQ_sum TCC_EA_WRREQ_64B_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmcm_$i
  ( cd $R && timeout -k 10 300 rocprofv3 --pmc $grp -d /tmp/pmcm_$i -o p --output-format csv -- python3 bench.py --steps 20 --warmup 5 --no-cpu-baseline "$@" > /tmp/pmcm_$i.log 2>&1 ) || { echo "pass $i ($grp) failed"; tail -3 /tmp/pmcm_$i.log; }
  f=$(find /tmp/pmcm_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: [0.0, 0])
for r in rows:
    if "csv_wave" in r["Kernel_Name"]:
        k = r["Counter_Name"]; acc[k][0] += float(r["Counter_Value"]); acc[k][1] += 1
for k, (v, n) in acc.items():
    print("%-28s per-launch %.6g  (launches %d)" % (k, v / n, n))
PY
done

#!/bin/bash
# Collects the rocprofv3 evidence for profiles/<name>/ on the GPU box:  tools/profile_round.sh <name>
# (kernel-trace stats and PMC passes are separate runs; --pmc is never combined with trace domains)
# usage: tools/profile_round.sh <name> ["extra bench.py arguments"]     e.g.  tools/profile_round.sh r02_C3 "--config C3"
name=${1:-r02_C2}
ARGS=${2:-}
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/$name
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# the config's own command: its iteration count after 100 warm-up iterations, result check, cpu_baseline and phases included (round 3:
# the lines of EVERY config carry cpu_baseline); the kernel trace below profiles the same command
CMD="python3 bench.py $ARGS"
# 1. un-profiled bench line (+ the driver's short command for the headline config)
( cd $R && timeout -k 10 600 $CMD > $out/bench_default.json 2> $out/bench_default.err ) || echo "bench failed"
[ -z "$ARGS" ] && ( cd $R && timeout -k 10 300 python3 bench.py --steps 20 --warmup 5 > $out/bench_driver_cmd.json 2> $out/bench_driver_cmd.err )
# 2. kernel trace + stats
rm -rf /tmp/kt
# (a process that issued a COOPERATIVE launch dies in exit() under rocprofv3 AFTER the tool has written its files -- profiles/README.md,
# tools/coop_exit_probe.hip: an empty kernel does the same -- so the exit status alone does not say "failed"; the whole log is kept)
( cd $R && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d /tmp/kt -o kt --output-format csv -- $CMD > /tmp/kt.log 2>&1 ) || echo "kernel-trace pass: exit status $? (see kernel_trace.log)"
cp /tmp/kt.log $out/kernel_trace.log
f=$(find /tmp/kt -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats.csv
# 2b. resident flows (C4): ONE launch of the configured length per kernel and nothing else (no warm-up iterations, no device prewarm, no
# phases), so that kernel_stats_single_launch.csv / iterations IS the time per iteration
case "$ARGS" in *C4*)
  rm -rf /tmp/kt1
  ( cd $R && timeout -k 10 600 rocprofv3 --kernel-trace --stats -d /tmp/kt1 -o kt --output-format csv -- python3 bench.py $ARGS --warmup 0 --prewarm-ms 0 --no-phases --no-cpu-baseline > /tmp/kt1.log 2>&1 ) || echo "single-launch kernel-trace pass: exit status $?"
  cp /tmp/kt1.log $out/kernel_trace_single_launch.log
  f=$(find /tmp/kt1 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp $f $out/kernel_stats_single_launch.csv
  grep -h "^{" /tmp/kt1.log > $out/bench_single_launch.json ;;
esac
# 3. PMC passes
python3 - > $out/pmc_summary.json <<'PY'
import json
print("{}")
PY
i=0
for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_INSTS_VALU" "SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_LDS SQ_INSTS_LDS" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM" "SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_ADD_F64"; do
  i=$((i+1)); rm -rf /tmp/pm_$i
  # (C4: --warmup 0 makes the counted csv_resident_kernel launch ONE launch of 60 iterations; the PM launch is the configured 1000 steps)
  W=40; case "$ARGS" in *C4*) W=0 ;; esac
  ( cd $R && timeout -k 10 300 rocprofv3 --pmc $grp -d /tmp/pm_$i -o p --output-format csv -- python3 bench.py $ARGS --steps 60 --warmup $W --no-cpu-baseline --no-phases --prewarm-ms 0 > /tmp/pm_$i.log 2>&1 ) || echo "pmc pass $i: exit status $? (see pmc_pass_$i.log)"
  cp /tmp/pm_$i.log $out/pmc_pass_$i.log
  f=$(find /tmp/pm_$i -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python3 - "$f" "$out/pmc_summary.json" <<'PY'
import csv, sys, json, collections
rows = list(csv.DictReader(open(sys.argv[1])))
acc = collections.defaultdict(lambda: collections.defaultdict(lambda: [0.0, 0]))
for r in rows:
    kn = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    a = acc[kn][r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
d = json.load(open(sys.argv[2]))
for kn, cs in acc.items():
    for c, (v, n) in cs.items():
        d.setdefault(kn, {})[c] = {"per_launch": v / n, "launches": n}
json.dump(d, open(sys.argv[2], "w"), indent=1, sort_keys=True)
PY
done
python3 - $out <<'PY'
import json, sys, csv
o = sys.argv[1]
b = json.loads(open(o + "/bench_default.json").read())
print("bench avg launch us", b["roofline"]["avg_launch_us"], "frac", b["roofline"]["frac"], "frac_wall", b["roofline"].get("frac_wall"), "kernel", b["roofline"]["kernel"],
      "checked", b.get("checked"), "cpu_baseline", (b.get("cpu_baseline") or {}).get("value"), "pm", (b.get("pm") or {}).get("us_per_step"))
try:
    for r in csv.DictReader(open(o + "/kernel_stats.csv")):
        print("kernel_stats:", r["Name"][:90], "calls", r["Calls"], "avg ns", r["AverageNs"])
except Exception as e: print("no kernel stats", e)
d = json.load(open(o + "/pmc_summary.json"))
for kn, cs in d.items():
    if "wave" in kn or "pm_" in kn or "resident" in kn:
        print(kn[:60], {k: round(v["per_launch"], 1) for k, v in cs.items()})
PY

#!/bin/bash
# Effective shader clock per kernel under the bench command: GRBM_GUI_ACTIVE (cycles the GPU was busy during the dispatch) / dispatch duration.
# One rocprofv3 --pmc pass.  Usage: tools/pmc_clock.sh <tag> [bench args]
tag=${1:-clk}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-extras "$@" > $O/pmc.log 2>&1
cd $R && python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
cc = sorted(glob.glob(O + "/pmc/**/*counter_collection.csv", recursive=True))
kt = sorted(glob.glob(O + "/pmc/**/*kernel_trace.csv", recursive=True))
dur = {}
for r in csv.DictReader(open(kt[0])):
    dur[r["Dispatch_Id"]] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), r["Kernel_Name"])
agg = collections.defaultdict(lambda: [0.0, 0.0, 0])
for r in csv.DictReader(open(cc[0])):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE": continue
    d = dur.get(r["Dispatch_Id"])
    if not d: continue
    k = d[1][:64]
    agg[k][0] += float(r["Counter_Value"]); agg[k][1] += d[0]; agg[k][2] += 1
with open(O + "/clock_summary.txt", "w") as out:
    for k, (cyc, ns, n) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:16]:
        line = f"{k:64s} calls {n:4d} avg {ns / n / 1e3:8.1f} us  GUI_ACTIVE/duration = {cyc / ns:6.3f} GHz-equivalent"
        print(line); out.write(line + "\n")
PY
find $O -name "*counter_collection.csv" -delete; find $O -name "*kernel_trace.csv" -delete

#!/bin/bash
# SQ counter pass over the bench command (one rocprofv3 --pmc pass, 8 SQ slots); summary per kernel -> gpurun_out/<tag>/sq_summary.txt
tag=${1:-sq}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $O/pmc -- python3 $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras > $O/pmc.log 2>&1
cd $R && python3 - "$O" <<'PY'
import csv, glob, sys, collections
O = sys.argv[1]
f = sorted(glob.glob(O + "/pmc/**/*counter_collection.csv", recursive=True))[0]
agg = collections.defaultdict(lambda: collections.defaultdict(float)); calls = collections.Counter()
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"][:70]
    agg[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "SQ_WAVE_CYCLES": calls[k] += 1
rows = sorted(agg.items(), key=lambda kv: -kv[1]["SQ_BUSY_CYCLES"])[:14]
with open(O + "/sq_summary.txt", "w") as out:
    for k, c in rows:
        wc = max(c["SQ_WAVE_CYCLES"], 1)
        line = (f"{k:70s} calls {calls[k]:4d} wait_any {c['SQ_WAIT_ANY']/wc:5.2f} wait_inst {c['SQ_WAIT_INST_ANY']/wc:5.2f} active {c['SQ_ACTIVE_INST_ANY']/wc:5.2f} "
                f"wait_lds {c['SQ_WAIT_INST_LDS']/wc:5.2f} mfma_busy/busy {c['SQ_VALU_MFMA_BUSY_CYCLES']/max(c['SQ_BUSY_CYCLES'],1):5.2f} bank_conf/wave_cyc {c['SQ_LDS_BANK_CONFLICT']/wc:6.3f}")
        print(line); out.write(line + "\n")
PY
find $O -name "*counter_collection.csv" -delete

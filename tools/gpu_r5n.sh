#!/bin/bash
# Round 5: vector-memory-path counters of conv_split_ad32_kernel<2,2,0> on 192->192 @34x60, product (abl0) against ablation 8 (cache-hot halo), one rocprofv3 --pmc pass per counter group.
tag=${1:-r05n}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R/tools/convbench; mkdir -p libs/abl0; cp $R/eagle_amd/libeagle_hip.so libs/abl0/
printf "3,1,192,192,34,60,50\n" > /tmp/l1.csv
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $O/counters_avail.txt 2>&1; grep -c . $O/counters_avail.txt
for grp in "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum" "TA_BUSY_avr TA_TA_BUSY_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_NC_READ_REQ_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_INST_CYCLES_VMEM"; do
  g=$(echo $grp | tr ' ' '_' | cut -c1-40)
  for m in 0 8; do
    LD_LIBRARY_PATH=$R/tools/convbench/libs/abl$m:$LD_LIBRARY_PATH TUNE_ONLY=21 rocprofv3 --pmc $grp --output-format csv -d $O/pmc_${m}_$g -- $R/tools/convbench/split_tune.out /tmp/l1.csv > $O/pmc_${m}_$g.log 2>&1
    python3 - "$O/pmc_${m}_$g" "$m" <<'PY'
import csv, glob, sys, collections
fs = sorted(glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True))
if not fs: print("abl" + sys.argv[2], "no counters (invalid name in the group?)"); sys.exit(0)
agg = collections.defaultdict(float); n = collections.Counter()
for r in csv.DictReader(open(fs[0])):
    if "ad32" in r["Kernel_Name"]:
        agg[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
print("abl" + sys.argv[2], {k: round(v / max(n[k], 1)) for k, v in agg.items()})
PY
  done
done 2>&1 | tee $O/pmc_summary.txt
find $O -name "*counter_collection.csv" -size +5M -delete

#!/bin/bash
# Developer probe: isolated fuse_sum launch — duration (kernel trace) and SQ / TA counters.  Usage (on the GPU box): tools/probes/fuse_probe.sh
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/fusepmc; mkdir -p $O; cd $R
for b in 20 50; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt$b -- python3 tools/probes/fuse_pmc.py $b > $O/kt$b.log 2>&1
  grep "algorithmic" $O/kt$b.log; grep fuse_sum $O/kt$b/*/*kernel_stats.csv | cut -d, -f1-4
done
for c in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU" "SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" "GRBM_GUI_ACTIVE TA_BUSY_avr TCP_PENDING_STALL_CYCLES_sum"; do
  n=$(echo $c | cut -d" " -f1)
  rocprofv3 --pmc $c --output-format csv -d $O/$n -- python3 tools/probes/fuse_pmc.py 50 > $O/$n.log 2>&1
done
python3 - <<PY
import csv,glob
for f in sorted(glob.glob("$O/SQ*/**/*counter_collection.csv", recursive=True))+sorted(glob.glob("$O/GRBM*/**/*counter_collection.csv", recursive=True)):
    for r in csv.DictReader(open(f)):
        if "fuse_sum" in r["Kernel_Name"]: print(r["Counter_Name"], r["Counter_Value"])
PY

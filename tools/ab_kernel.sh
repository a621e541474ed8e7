#!/bin/bash
# same-box A/B of ONE kernel's average duration (rocprofv3 --stats) over the cadence bench: tools/ab_kernel.sh <kernel-name-substring>
K=$1; R=$GRAFT_REPO_ROOT
for v in ${VARIANTS:-A B A B}; do
  cp $R/tools/convbench/lib$v.so $R/eagle_amd/libeagle_hip.so
  rm -rf /tmp/abk; (cd /tmp && TMPDIR=/tmp rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abk -- python3 $R/bench.py --steps 4 --warmup 1 --no-cpu-baseline --cadence 25 > /dev/null 2>&1)
  echo -n "$v "; grep "$K" /tmp/abk/*/*kernel_stats.csv | awk -F, '{gsub(/"/,""); print $1, "calls", $2, "avg_us", $4/1000}' | cut -c1-120
done

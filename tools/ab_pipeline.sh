#!/bin/bash
# same-box A/B of library builds in the whole pipeline: tools/convbench/libs/<name>/libeagle_hip.so, alternating; usage: tools/ab_pipeline.sh "old new" [reps] [bench args]
V=${1:-"old new"}; R=${2:-3}; shift 2
for rep in $(seq $R); do for v in $V; do
  EAGLE_HIP_LIB=$PWD/tools/convbench/libs/$v/libeagle_hip.so python bench.py --no-cpu-baseline --no-extras "$@" 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v', j['value'], 'fps  conv', j['roofline']['conv_ms_per_step'], 'ms/step  dominant', j['roofline']['dominant_kernel']['avg_us'], 'us')"
done; done

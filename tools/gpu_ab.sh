#!/bin/bash
# Same-box A/B of a library switch: alternates `<ENV>=0 python bench.py` and the default.  Usage: tools/gpu_ab.sh <tag> <ENV_NAME> [bench args]
#   e.g. tools/gpu_ab.sh ad EAGLE_CONV_AD --all-layers        (switches: EAGLE_CONV_AD, EAGLE_CONV_AD2, EAGLE_NO_FUSED_ARGMAX=1 ...)
tag=${1:-ab}; envn=${2:-EAGLE_CONV_AD}; shift; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
for i in 1 2; do
  env $envn=0 timeout 600 python bench.py --no-cpu-baseline --no-extras "$@" > $O/bench_off$i.json 2>> $O/bench.err
  timeout 600 python bench.py --no-cpu-baseline --no-extras "$@" > $O/bench_on$i.json 2>> $O/bench.err
done
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    j=json.load(open(f)); r=j["roofline"]
    print(os.path.basename(f), j["value"], "frames/s  conv", r["achieved"], "TFLOP/s", r["conv_ms_per_step"], "ms/step")
PY

#!/bin/bash
# Round 5: non-temporal stores / residual loads over the WHOLE split family (libs/abl41: -DEAGLE_STORE_NT=1; libs/abl42: + -DEAGLE_RES_NT=1) against the product (abl0), whole
# pipeline, three alternating triples, with the per-layer table of the biggest classes.
tag=${1:-r05l}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R/tools/convbench; mkdir -p libs/abl0; cp $R/eagle_amd/libeagle_hip.so libs/abl0/
cd $R
for i in 1 2 3; do for m in 0 41 42; do
  EAGLE_HIP_LIB=$R/tools/convbench/libs/abl$m/libeagle_hip.so timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --all-layers > $O/pipe_nt_${m}_$i.json 2> $O/pipe_nt_${m}_$i.err
  python3 - <<PY
import json
d = json.loads(open("$O/pipe_nt_${m}_$i.json").readline())
rows = {r["layer"][:24]: r["avg_us"] for r in d.get("roofline_conv_layers", [])[:9]}
print("pair $i lib abl$m", d["value"], "conv ms", d["roofline"]["conv_ms_per_step"], "frac", d["roofline"]["frac"], rows)
PY
done; done

#!/bin/bash
# Round 5: halo slab requests spread over the chunk's taps (libs/abl1 built with -DEAGLE_M32_HSPREAD=1) against the product (abl0): isolated and through the pipeline.
tag=${1:-r05i}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R/tools/convbench
mkdir -p libs/abl0; cp $R/eagle_amd/libeagle_hip.so libs/abl0/
LAYER=3,1,192,192,34,60,50 ONLY=21 ABLS="0 1" ./ablate_split.sh run 3 > $O/hspread_192.txt 2>&1; cat $O/hspread_192.txt
LAYER=3,1,96,96,68,120,50 ONLY=24 ABLS="0 1" ./ablate_split.sh run 3 > $O/hspread_96.txt 2>&1; cat $O/hspread_96.txt
LAYER=3,1,384,384,17,30,50 ONLY=21 ABLS="0 1" ./ablate_split.sh run 3 > $O/hspread_384.txt 2>&1; cat $O/hspread_384.txt
cd $R
for i in 1 2 3; do for m in 0 1; do
  EAGLE_HIP_LIB=$R/tools/convbench/libs/abl$m/libeagle_hip.so timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --all-layers > $O/pipe_hs_${m}_$i.json 2> $O/pipe_hs_${m}_$i.err
  python3 - <<PY
import json
d = json.loads(open("$O/pipe_hs_${m}_$i.json").readline())
rows = {r["layer"]: r["avg_us"] for r in d.get("roofline_conv_layers", []) if "3x3/1" in r["layer"] and any(k in r["layer"] for k in ("96->96", "192->192", "384->384"))}
print("pair $i HSPREAD=$m", d["value"], "conv ms", d["roofline"]["conv_ms_per_step"], "frac", d["roofline"]["frac"], rows)
PY
done; done

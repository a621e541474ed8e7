#!/bin/bash
# Round 5, sixth GPU pass: exclusive graph capture (the two tests that still failed), the side-by-side tile forms of the 32x32x16 kernel (variants 23 / 24): parity,
# isolated timing against 8 / 9 / 21 / 22, pipeline EAGLE_CONV_M32 = 0 / 1 / 4.   Usage: tools/gpu_r5f.sh <tag>
tag=${1:-r05f}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_edges.py tests/test_gpu_flow.py -m gpu -q > $O/edges_flow.log 2>&1; tail -2 $O/edges_flow.log; grep -E "^FAILED|^ERROR" $O/edges_flow.log | head
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "a_direct_m32" > $O/m32_parity.log 2>&1; tail -2 $O/m32_parity.log; grep -E "^FAILED|^ERROR" $O/m32_parity.log | head
cd $R/tools/convbench
printf "3,1,96,96,68,120,50\n3,1,192,192,34,60,50\n" > /tmp/l64.csv
for r in 1 2 3; do for v in 8 21 23 9 22 24; do
  TUNE_ONLY=$v ./split_tune.out /tmp/l64.csv 2>/dev/null
done; done | awk -F, '{k=$4"->"$5"@"$6"x"$7" v"$12" res"$13; s[k]+=$14; c[k]++; if(!(k in m)||$14<m[k]) m[k]=$14} END {for (k in s) printf "%s mean %.1f min %.1f us\n", k, s[k]/c[k], m[k]}' | sort > $O/m32_w64_isolated.txt
cat $O/m32_w64_isolated.txt
cd $R
for i in 1 2 3; do for m in 0 1 4; do
  EAGLE_CONV_M32=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --all-layers > $O/pipe_m32_${m}_$i.json 2> $O/pipe_m32_${m}_$i.err
  python3 - <<PY
import json
d = json.loads(open("$O/pipe_m32_${m}_$i.json").readline())
rows = {r["layer"]: r["avg_us"] for r in d.get("roofline_conv_layers", []) if "3x3/1" in r["layer"] and any(k in r["layer"] for k in ("96->96", "192->192", "384->384"))}
print("pair $i M32=$m", d["value"], "conv ms", d["roofline"]["conv_ms_per_step"], "frac", d["roofline"]["frac"], rows)
PY
done; done

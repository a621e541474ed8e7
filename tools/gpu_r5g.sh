#!/bin/bash
# Round 5, seventh GPU pass: variant 25 (three halo slots, two chunks' requests together): parity, isolated against 8 / 21, pipeline EAGLE_CONV_M32 = 0 / 6 / 5.  Usage: tools/gpu_r5g.sh <tag>
tag=${1:-r05g}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "a_direct_m32 and halo3" > $O/m32_parity_h3.log 2>&1; tail -2 $O/m32_parity_h3.log; grep -E "^FAILED|^ERROR" $O/m32_parity_h3.log | head
cd $R/tools/convbench
printf "3,1,192,192,34,60,50\n3,1,384,384,17,30,50\n" > /tmp/lh3.csv
for r in 1 2 3; do for v in 8 21 25; do
  TUNE_ONLY=$v ./split_tune.out /tmp/lh3.csv 2>/dev/null
done; done | awk -F, '{k=$4"->"$5"@"$6"x"$7" v"$12" res"$13; s[k]+=$14; c[k]++; if(!(k in m)||$14<m[k]) m[k]=$14} END {for (k in s) printf "%s mean %.1f min %.1f us\n", k, s[k]/c[k], m[k]}' | sort > $O/m32_h3_isolated.txt
cat $O/m32_h3_isolated.txt
cd $R
for i in 1 2 3; do for m in 0 6 5; do
  EAGLE_CONV_M32=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --all-layers > $O/pipe_m32_${m}_$i.json 2> $O/pipe_m32_${m}_$i.err
  python3 - <<PY
import json
d = json.loads(open("$O/pipe_m32_${m}_$i.json").readline())
rows = {r["layer"]: r["avg_us"] for r in d.get("roofline_conv_layers", []) if "3x3/1" in r["layer"] and any(k in r["layer"] for k in ("96->96", "192->192", "384->384"))}
print("pair $i M32=$m", d["value"], "conv ms", d["roofline"]["conv_ms_per_step"], "frac", d["roofline"]["frac"], rows)
PY
done; done

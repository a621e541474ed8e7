#!/bin/bash
# Round 5: (1) variant 26 (whole-line halo requests, two chunks per request): parity, isolated against 8 / 21; (2) nt stores (abl32) and nt stores + nt residual loads (abl6)
# isolated and through the pipeline against the product (abl0 = this build).
tag=${1:-r05k}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -q -k "a_direct_m32 and lines" > $O/m32_parity_k2.log 2>&1; tail -2 $O/m32_parity_k2.log; grep -E "^FAILED|^ERROR" $O/m32_parity_k2.log | head
cd $R/tools/convbench
printf "3,1,192,192,34,60,50\n3,1,384,384,17,30,50\n" > /tmp/lk2.csv
for r in 1 2 3; do for v in 8 21 26; do
  TUNE_ONLY=$v ./split_tune.out /tmp/lk2.csv 2>/dev/null
done; done | awk -F, '{k=$4"->"$5"@"$6"x"$7" v"$12" res"$13; s[k]+=$14; c[k]++; if(!(k in m)||$14<m[k]) m[k]=$14} END {for (k in s) printf "%s mean %.1f min %.1f us\n", k, s[k]/c[k], m[k]}' | sort > $O/m32_k2_isolated.txt
cat $O/m32_k2_isolated.txt
mkdir -p libs/abl0; cp $R/eagle_amd/libeagle_hip.so libs/abl0/
LAYER=3,1,96,96,68,120,50 ONLY=24 ABLS="0 32 6" ./ablate_split.sh run 3 > $O/nt_store_96.txt 2>&1; cat $O/nt_store_96.txt
LAYER=3,1,192,192,34,60,50 ONLY=21 ABLS="0 32 6" ./ablate_split.sh run 3 > $O/nt_store_192.txt 2>&1; cat $O/nt_store_192.txt
cd $R
for i in 1 2 3; do for m in 0 32 6; do
  EAGLE_HIP_LIB=$R/tools/convbench/libs/abl$m/libeagle_hip.so timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --all-layers > $O/pipe_nt_${m}_$i.json 2> $O/pipe_nt_${m}_$i.err
  python3 - <<PY
import json
d = json.loads(open("$O/pipe_nt_${m}_$i.json").readline())
rows = {r["layer"]: r["avg_us"] for r in d.get("roofline_conv_layers", []) if "3x3/1" in r["layer"] and any(k in r["layer"] for k in ("96->96", "192->192", "384->384"))}
print("pair $i lib abl$m", d["value"], "conv ms", d["roofline"]["conv_ms_per_step"], "frac", d["roofline"]["frac"], rows)
PY
done; done
for i in 1 2; do for m in 6 7; do
  EAGLE_CONV_M32=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --all-layers > $O/pipe_k2_${m}_$i.json 2> $O/pipe_k2_${m}_$i.err
  python3 - <<PY
import json
d = json.loads(open("$O/pipe_k2_${m}_$i.json").readline())
rows = {r["layer"]: r["avg_us"] for r in d.get("roofline_conv_layers", []) if "3x3/1" in r["layer"] and any(k in r["layer"] for k in ("96->96", "192->192", "384->384"))}
print("pair $i M32=$m", d["value"], "conv ms", d["roofline"]["conv_ms_per_step"], "frac", d["roofline"]["frac"], rows)
PY
done; done

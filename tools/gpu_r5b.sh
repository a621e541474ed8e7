#!/bin/bash
# Round 5: the 32x32x16 form of the split A-direct kernel (variants 21 / 22).  Parity matrix, isolated same-box timing against variants 8 / 9 (alternating
# sweeps of tools/convbench/split_tune.out, which links the built library), then the whole pipeline with EAGLE_CONV_M32=0/1 alternating.  Usage: tools/gpu_r5b.sh <tag> [pairs]
tag=${1:-r05b}; PAIRS=${2:-3}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
# (r05a: the full-size multirank composition printed its timed region and then died without a JSON line: exit code and a faulthandler trace first)
timeout 600 python -X faulthandler bench.py --force-multirank-path --backend nccl --gather rccl --steps 20 --warmup 5 --no-cpu-baseline > $O/mr_debug.json 2> $O/mr_debug.err; echo "multirank rc=$?"; tail -30 $O/mr_debug.err; head -c 300 $O/mr_debug.json
timeout 1200 python -m pytest tests/test_gpu_ops.py -m gpu -q -x -k "a_direct_m32" > $O/m32_parity.log 2>&1; tail -3 $O/m32_parity.log
cd $R/tools/convbench
for r in 1 2 3; do for v in 8 21 9 22; do
  TUNE_ONLY=$v ./split_tune.out layers_m32.csv 2>/dev/null
done; done | awk -F, '{k=$4"->"$5"@"$6"x"$7" v"$12" res"$13; s[k]+=$14; c[k]++; if(!(k in m)||$14<m[k]) m[k]=$14} END {for (k in s) printf "%s mean %.1f min %.1f us\n", k, s[k]/c[k], m[k]}' | sort > $O/m32_isolated.txt
cat $O/m32_isolated.txt
cd $R
for i in $(seq $PAIRS); do for m in 0 1; do
  EAGLE_CONV_M32=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --all-layers > $O/pipe_m32_${m}_$i.json 2> $O/pipe_m32_${m}_$i.err
  python3 - <<PY
import json
d = json.loads(open("$O/pipe_m32_${m}_$i.json").read().strip().splitlines()[-1])
rows = {r["layer"]: r["avg_us"] for r in d.get("roofline_conv_layers", []) if "3x3/1" in r["layer"] and any(k in r["layer"] for k in ("96->96", "192->192", "384->384"))}
print("pair $i M32=$m", d["value"], "conv ms", d["roofline"]["conv_ms_per_step"], "frac", d["roofline"]["frac"], rows)
PY
done; done
timeout 900 python bench.py --latency-only --latency-calls 100 > $O/latency_modes.json 2> $O/latency_modes.err; grep "latency B" $O/latency_modes.err

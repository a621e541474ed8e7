#!/bin/bash
# Round 5 (record; the "48NR=12" arm needs the one-line conv.hip change quoted in profiles/r05ab_*): the Cout = 48 A-direct forms re-measured through the pipeline now that the stores are non-temporal: base (variant 18 generic 8 x 48 tiles) against EAGLE_CONV_KQ=12 / 13 (all
# 48->48 launches on the K-split / the 16 x 32 single-buffer form) and EAGLE_CONV_48NR=12 (K-split for the residual-free launches only), three alternating rounds.
tag=${1:-r05ab}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
for i in 1 2 3; do for m in "base" "EAGLE_CONV_KQ=12" "EAGLE_CONV_KQ=13" "EAGLE_CONV_48NR=12"; do
  if [ "$m" = base ]; then e=""; else e="$m"; fi
  env $e timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --all-layers > $O/p_${m}_$i.json 2> $O/p_${m}_$i.err
  python3 - <<PY
import json
d = json.loads(open("$O/p_${m}_$i.json").readline())
rows = {r["layer"]: r["avg_us"] for r in d.get("roofline_conv_layers", []) if "48->48 @135" in r["layer"]}
print("round $i $m", d["value"], "conv ms", d["roofline"]["conv_ms_per_step"], "frac", d["roofline"]["frac"], rows)
PY
done; done

#!/bin/bash
# Round 5: the product (nt stores over the split family + nt residual loads in the 32x32x16 kernels: the default now) against the same with fuse_sum's output stores non-temporal (abl43),
# three alternating pairs; then the GPU suite on the product.
tag=${1:-r05m}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R/tools/convbench; mkdir -p libs/abl0; cp $R/eagle_amd/libeagle_hip.so libs/abl0/
cd $R
for i in 1 2 3; do for m in 0 43; do
  EAGLE_HIP_LIB=$R/tools/convbench/libs/abl$m/libeagle_hip.so timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --all-layers > $O/pipe_${m}_$i.json 2> $O/pipe_${m}_$i.err
  python3 - <<PY
import json
d = json.loads(open("$O/pipe_${m}_$i.json").readline())
print("pair $i lib abl$m", d["value"], "conv ms", d["roofline"]["conv_ms_per_step"], "frac", d["roofline"]["frac"], [(r["kernel"], r["avg_us"]) for r in d.get("roofline_hbm", []) if r["kernel"] in ("fuse_sum", "preprocess")])
PY
done; done
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log | tail -1; grep -E "^FAILED|^ERROR" $O/gpu_tests.log | head

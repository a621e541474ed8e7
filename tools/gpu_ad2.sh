#!/bin/bash
tag=${1:-ad2}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x > $O/ops.log 2>&1; tail -2 $O/ops.log
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_on.json 2> $O/bench.err
EAGLE_CONV_AD=0 timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_off.json 2>> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_on2.json 2>> $O/bench.err
timeout 900 python bench.py --detector l --imgsz 960 --height 1080 --width 1920 --batch 25 --steps 40 --no-cpu-baseline > $O/bench_cfg3.json 2>> $O/bench.err
timeout 1200 python -m pytest tests/test_gpu_pipeline.py -q -x > $O/pipe.log 2>&1; grep -E "passed|failed" $O/pipe.log | tail -1
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    j=json.load(open(f)); r=j["roofline"]
    print(os.path.basename(f), j["value"], "fps  conv", r["achieved"], "TF", r["conv_ms_per_step"], "ms", [ (l["layer"].split(" ")[1], l["avg_us"]) for l in j["roofline_conv_layers"][:5]])
PY

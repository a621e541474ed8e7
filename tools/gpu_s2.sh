#!/bin/bash
tag=${1:-s2}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "a_direct" > $O/ops.log 2>&1; tail -4 $O/ops.log
EAGLE_CONV_AD2=0 timeout 600 python bench.py --no-cpu-baseline --no-extras --all-layers > $O/bench_off.json 2> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --no-extras --all-layers > $O/bench_on.json 2>> $O/bench.err
EAGLE_CONV_AD2=0 timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_off2.json 2>> $O/bench.err
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_on2.json 2>> $O/bench.err
python - <<PY
import json,glob,os
for f in sorted(glob.glob("$O/bench_*.json")):
    try:
        j=json.load(open(f)); r=j["roofline"]
        print(os.path.basename(f), j["value"], "fps  conv", r["achieved"], "TF", r["conv_ms_per_step"], "ms")
        if "all" or 1:
            for l in j["roofline_conv_layers"]:
                if l["layer"].startswith("3x3/2"): print("    ", l["layer"], l["launches_per_step"], l["avg_us"], "us", l["TFLOPs"], "TF")
    except Exception as e: print(f, e)
PY

#!/bin/bash
# A-direct conv variant: parity tests, fp16 pipeline parity, bench A/B (EAGLE_CONV_AD=0 vs default).  Usage: tools/gpu_ad_check.sh <tag>
tag=${1:-ad}; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O; cd $R
timeout 900 python -m pytest tests/test_gpu_ops.py -q -x -k "a_direct or weight_stationary or conv" > $O/ops.log 2>&1; tail -3 $O/ops.log
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_edges.py -q -x > $O/pipe.log 2>&1; tail -3 $O/pipe.log
EAGLE_CONV_AD=0 timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_off.json 2> $O/bench_off.err
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_on.json 2> $O/bench_on.err
EAGLE_CONV_AD=0 timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_off2.json 2>> $O/bench_off.err
timeout 600 python bench.py --no-cpu-baseline --no-extras > $O/bench_on2.json 2>> $O/bench_on.err
python - <<'PY'
import json,glob,os
O=os.environ.get("GRAFT_REPO_ROOT",".")+"/gpurun_out/"+os.environ.get("ADTAGX","ad1")
for f in sorted(glob.glob(O+"/bench_*.json")):
    try:
        j=json.load(open(f)); r=j["roofline"]
        print(os.path.basename(f), j["value"], "fps  conv", r["achieved"], "TF  conv_ms", r["conv_ms_per_step"])
        if "on" in f:
            for l in j.get("roofline_conv_layers",[])[:6]: print("   ", l["layer"], l["avg_us"], "us", l["TFLOPs"], "TF")
    except Exception as e: print(f, "unreadable", e)
PY

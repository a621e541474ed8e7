#!/bin/bash
# Round 5: the blocked NMS sweep: detection parity tests, the kernel's time in the bench (roofline_hbm 'nms'), B = 1 latency; then EAGLE_CONV_48NR=0/1 pairs.
tag=${1:-r05y}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 1800 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ops.py -m gpu -q -x -k "not conv_ and not fuse and not preprocess" > $O/tests_nms.log 2>&1; tail -2 $O/tests_nms.log; grep -E "^FAILED|^ERROR" $O/tests_nms.log | head
timeout 900 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --exact-frames 0 --fast-frames 0 --cfg3-frames 0 --latency-calls 100 > $O/bench_nms.json 2> $O/bench_nms.err
python3 -c "
import json; d=json.loads(open('$O/bench_nms.json').readline()); print(d['value'], [(r['kernel'], r['avg_us']) for r in d['roofline_hbm'] if r['kernel'] in ('nms','yolo_decode')]); [print(r['frames_per_call'], r['mode'][:12], r['median_ms']) for r in d['latency']['rows']]; print(d['parity_counters'][d['parity_counters']['default']] if 'parity_counters' in d else None)"
for i in 1 2 3; do for m in 0 1; do EAGLE_CONV_48NR=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/nr48_${m}_$i.json 2> $O/nr48_${m}_$i.err; echo "pair $i 48NR=$m $(grep -o 'timed region.*' $O/nr48_${m}_$i.err)"; done; done

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6k
L=gpurun_out/r6k/bneck_pipeline_ab.log
: > $L
for rep in 1 2 3; do
  for f in 0 1; do
    echo "EAGLE_BNECK_FUSED=$f rep $rep" >> $L
    EAGLE_BNECK_FUSED=$f timeout 600 python bench.py --no-extras --no-cpu-baseline --latency-calls 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'frames/s', d['ms_per_step'], 'ms/step; conv family', d['roofline']['conv_ms_per_step'], 'ms, frac', d['roofline']['frac'])" >> $L 2>&1
  done
done
cat $L
EAGLE_BNECK_FUSED=1 timeout 600 python bench.py --no-extras --no-cpu-baseline --latency-calls 0 --all-layers 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in d['roofline_conv_layers']:
    if '135x240' in r['layer'] or 'bneck' in r['layer']: print(r['layer'], r['launches_per_step'], r['avg_us'], r['ms_per_step'])
" > gpurun_out/r6k/layers_fused.log 2>&1; cat gpurun_out/r6k/layers_fused.log
EAGLE_BNECK_FUSED=0 timeout 600 python bench.py --no-extras --no-cpu-baseline --latency-calls 0 --all-layers 2>/dev/null | python -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in d['roofline_conv_layers']:
    if '135x240' in r['layer'] or 'bneck' in r['layer']: print(r['layer'], r['launches_per_step'], r['avg_us'], r['ms_per_step'])
" > gpurun_out/r6k/layers_unfused.log 2>&1; cat gpurun_out/r6k/layers_unfused.log

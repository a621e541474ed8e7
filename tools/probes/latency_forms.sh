cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6s
for f in 0 1; do echo "EAGLE_BNECK_FORM=$f" >> gpurun_out/r6s/latency_forms.log; EAGLE_BNECK_FORM=$f LATENCY_BATCHES=1,2,4 timeout 600 python bench.py --latency-only --latency-calls 100 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in d['latency']:
    if r['mode']=='multi_stream+graph': print(r['frames_per_call'], r['median_ms'], r['p99_ms'])
" >> gpurun_out/r6s/latency_forms.log; done
echo "EAGLE_BNECK_FUSED=0" >> gpurun_out/r6s/latency_forms.log; EAGLE_BNECK_FUSED=0 LATENCY_BATCHES=1,2,4 timeout 600 python bench.py --latency-only --latency-calls 100 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in d['latency']:
    if r['mode']=='multi_stream+graph': print(r['frames_per_call'], r['median_ms'], r['p99_ms'])
" >> gpurun_out/r6s/latency_forms.log
cat gpurun_out/r6s/latency_forms.log

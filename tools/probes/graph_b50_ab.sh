cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6an; L=gpurun_out/r6an/graph_b50_ab.log; : > $L
for rep in 1 2 3; do for f in "" "--graph"; do echo "bench.py $f rep $rep" >> $L
  timeout 600 python bench.py $f --no-extras --no-cpu-baseline --latency-calls 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'frames/s', d['ms_per_step'], 'ms/step')" >> $L 2>&1
done; done
LATENCY_BATCHES=32,50 python bench.py --latency-only --latency-calls 40 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
for r in d['latency']:
    if 'multi' in r['mode']: print(r['frames_per_call'], r['mode'], r['median_ms'], round(r['frames_per_call']/r['median_ms']*1000,1))
" >> $L
cat $L

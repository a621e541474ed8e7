# one-frame call latency with an environment knob on / off: bash tools/probes/latency_env_ab.sh <tag> VAR=value [VAR2=value ...]   (A/B/A/B; default mode = multi_stream+graph)
cd $GRAFT_REPO_ROOT; tag=$1; shift; mkdir -p gpurun_out/$tag; L=gpurun_out/$tag/latency_env_ab.log
one() { LATENCY_BATCHES=1,2,4 timeout 600 "$@" python bench.py --latency-only --latency-calls 100 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  '.join('B=%d median %.3f ms p99 %.3f' % (r['frames_per_call'], r['median_ms'], r['p99_ms']) for r in d['latency'] if r['mode']=='multi_stream+graph'))
"; }
for rep in 1 2; do
  echo "baseline:" >> $L; one env >> $L
  echo "$*:" >> $L; one env "$@" >> $L
done
cat $L

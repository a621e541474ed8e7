cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6w; L=gpurun_out/r6w/latency_ring_b8_16.log
one() { LATENCY_BATCHES=8,12,16 timeout 600 "$@" python bench.py --latency-only --latency-calls 60 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('  '.join('B=%d %s median %.3f ms' % (r['frames_per_call'], r['mode'], r['median_ms']) for r in d['latency'] if r['mode'] in ('multi_stream+graph','multi_stream')))
"; }
for rep in 1 2; do
  for v in default 6 3; do echo "ring $v:" >> $L; if [ $v = default ]; then one env >> $L; else one env EAGLE_CONV_M32_RING=$v >> $L; fi; done
done
cat $L

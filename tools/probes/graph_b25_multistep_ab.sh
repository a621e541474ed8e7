# graph replay against plain launches in MULTI-step calls at 16 / 25 / 32 frames per step (bench.py --batch B --steps 40 [--graph]), two alternating pairs each
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r6ao; L=gpurun_out/r6ao/graph_multistep_ab.log; : > $L
for B in 16 25 32; do for rep in 1 2; do for f in "" "--graph"; do echo "batch $B $f rep $rep" >> $L
  timeout 600 python bench.py --batch $B --steps 40 --warmup 4 $f --no-extras --no-cpu-baseline --latency-calls 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'frames/s', d['ms_per_step'], 'ms/step')" >> $L 2>&1
done; done; done
cat $L

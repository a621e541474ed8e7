cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6q
timeout 2400 python -m pytest tests -x -q -m gpu > gpurun_out/r6q/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6q/gpu_tests.log; tail -4 gpurun_out/r6q/gpu_tests.log
for t in 16 32 64 128; do echo "threads $t" >> gpurun_out/r6q/cpu_baseline_thread_sweep.txt; timeout 600 python bench.py --cpu-baseline-only --cpu-threads $t --cpu-frames 40 2>/dev/null | tail -1 >> gpurun_out/r6q/cpu_baseline_thread_sweep.txt; done
cat gpurun_out/r6q/cpu_baseline_thread_sweep.txt
timeout 1500 python bench.py > gpurun_out/r6q/bench_default_1gpu.json 2> gpurun_out/r6q/bench_default_1gpu.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r6q/bench_default_1gpu.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['dominant_kernel'])
print('realistic', d.get('realistic'))
print('resident', d.get('resident',{}).get('value'), 'exact', d.get('exact_family',{}).get('value'), 'cfg3', {k:(v.get('value') if isinstance(v,dict) else v) for k,v in d.get('cfg3',{}).items()})
print('parity', d.get('parity_counters'))
print('cpu', d.get('cpu_baseline'))
for r in d.get('latency',{}).get('rows',[]): print(r)
PY

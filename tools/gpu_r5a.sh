#!/bin/bash
# Round 5, first GPU pass: the new boundary / runtime-composition tests, the whole suite, and the same-box alternating A/B of the two process
# compositions (torch-free default against torch-first nccl + library RCCL at world 1; VERDICT r4 task 1b).  Usage: tools/gpu_r5a.sh <tag>
tag=${1:-r05a}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_edges.py -m gpu -q -x -k "strided or bad_strides or multirank_bench_uses or torch_after" > $O/new_tests.log 2>&1; tail -3 $O/new_tests.log
for i in 1 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/ab_default_$i.json 2> $O/ab_default_$i.err
  timeout 600 python bench.py --force-multirank-path --backend nccl --gather rccl --steps 20 --warmup 5 --no-cpu-baseline > $O/ab_multirank_$i.json 2> $O/ab_multirank_$i.err
  python3 - <<PY
import json
for n in ("default", "multirank"):
    try:
        d = json.loads(open("$O/ab_%s_$i.json" % n).read().strip().splitlines()[-1])
        print("pair $i", n, d["value"], d["ms_per_step"], d["config"]["gather"], d["config"].get("cpu_binding_rank0"))
    except Exception as e:
        print("pair $i", n, "FAILED", e)
PY
done
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log | tail -1

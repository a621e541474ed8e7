#!/bin/bash
# Round 5, fifth GPU pass: the suite after the thread-local capture fix; ablation 8 (cache-hot non-zero halo) next to 0 and 3; the torch-bound composition with and without
# hipGraph replay; the suite's conv / pipeline tests and three more pipeline pairs with EAGLE_CONV_M32=1.   Usage: tools/gpu_r5e.sh <tag>
tag=${1:-r05e}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log | tail -1; grep -E "^FAILED|^ERROR" $O/gpu_tests.log | head
cd $R/tools/convbench
cp $R/eagle_amd/libeagle_hip.so libs/abl0/
LAYER=3,1,192,192,34,60,50 ONLY=21 ABLS="0 3 8" ./ablate_split.sh run 3 > $O/m32_ablation8_192.txt 2>&1; cat $O/m32_ablation8_192.txt
cd $R
for i in 1 2; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/ab_default_$i.json 2> $O/ab_default_$i.err; echo "default $(grep -o 'timed region.*' $O/ab_default_$i.err)"
  timeout 600 python bench.py --force-multirank-path --backend nccl --gather rccl --steps 20 --warmup 5 --no-cpu-baseline > $O/ab_multirank_$i.json 2> $O/ab_multirank_$i.err; echo "multirank $(grep -o 'timed region.*' $O/ab_multirank_$i.err | tr '\n' ' ')"
  timeout 600 python bench.py --force-multirank-path --backend nccl --gather rccl --steps 20 --warmup 5 --no-cpu-baseline --graph > $O/ab_multirank_graph_$i.json 2> $O/ab_multirank_graph_$i.err; echo "multirank+graph $(grep -o 'timed region.*' $O/ab_multirank_graph_$i.err | tr '\n' ' ')"
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --graph > $O/ab_default_graph_$i.json 2> $O/ab_default_graph_$i.err; echo "default+graph $(grep -o 'timed region.*' $O/ab_default_graph_$i.err)"
done
EAGLE_CONV_M32=1 timeout 1800 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_ops.py -m gpu -q > $O/gpu_tests_m32.log 2>&1; echo "M32=1: $(grep -E 'passed|failed' $O/gpu_tests_m32.log | tail -1)"; grep -E "^FAILED|^ERROR" $O/gpu_tests_m32.log | head
for i in 1 2 3; do for m in 0 1; do
  EAGLE_CONV_M32=$m timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/pipe_m32_${m}_$i.json 2> $O/pipe_m32_${m}_$i.err
  echo "pair $i M32=$m $(grep -o 'timed region.*' $O/pipe_m32_${m}_$i.err)"
done; done

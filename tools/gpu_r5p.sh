#!/bin/bash
# Round 5: (1) graph replay inside long calls (use_graph auto = 2 for batch > 8): the tests that touch it, then default against --no-graph, alternating; (2) the exact family's stores
# non-temporal (libs/abl44: -DEAGLE_F32_STORE_NT=1) against the product on the exact family and on cfg 3's default handle.
tag=${1:-r05p}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_edges.py -m gpu -q -x -k "small_batch or graph or batch_and_position or property or properties or multirank or bench_default or strided or host_fed" > $O/tests_graph.log 2>&1; tail -2 $O/tests_graph.log; grep -E "^FAILED|^ERROR" $O/tests_graph.log | head
for i in 1 2 3; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/g_auto_$i.json 2> $O/g_auto_$i.err; echo "auto(graph) $(grep -o 'timed region.*' $O/g_auto_$i.err)"
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --no-graph > $O/g_plain_$i.json 2> $O/g_plain_$i.err; echo "plain $(grep -o 'timed region.*' $O/g_plain_$i.err)"
done
cd $R/tools/convbench; mkdir -p libs/abl0; cp $R/eagle_amd/libeagle_hip.so libs/abl0/; cd $R
for i in 1 2; do for m in 0 44; do
  EAGLE_HIP_LIB=$R/tools/convbench/libs/abl$m/libeagle_hip.so timeout 900 python bench.py --precision f32 --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/f32_${m}_$i.json 2> $O/f32_${m}_$i.err; echo "exact family lib abl$m $(grep -o 'timed region.*' $O/f32_${m}_$i.err)"
  EAGLE_HIP_LIB=$R/tools/convbench/libs/abl$m/libeagle_hip.so timeout 900 python bench.py --height 1080 --width 1920 --detector l --imgsz 960 --batch 25 --steps 12 --warmup 2 --no-extras --no-cpu-baseline > $O/cfg3_${m}_$i.json 2> $O/cfg3_${m}_$i.err; echo "cfg3 default lib abl$m $(grep -o 'timed region.*' $O/cfg3_${m}_$i.err)"
done; done

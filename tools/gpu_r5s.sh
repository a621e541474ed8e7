#!/bin/bash
# Round 5: does the graph replay inside long calls (use_graph = 2) cost the exact family / cfg 3 anything?  auto against --no-graph, alternating, product library (nt stores on).
tag=${1:-r05s}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
for i in 1 2; do for g in "" "--no-graph"; do
  timeout 900 python bench.py --precision f32 --steps 20 --warmup 3 --no-extras --no-cpu-baseline $g > $O/f32_${i}_x$g.json 2> $O/f32_${i}_x$g.err; echo "exact family [$g] $(grep -o 'timed region.*' $O/f32_${i}_x$g.err)"
  timeout 900 python bench.py --height 1080 --width 1920 --detector l --imgsz 960 --batch 25 --steps 20 --warmup 3 --no-extras --no-cpu-baseline $g > $O/cfg3_${i}_x$g.json 2> $O/cfg3_${i}_x$g.err; echo "cfg3 default [$g] $(grep -o 'timed region.*' $O/cfg3_${i}_x$g.err)"
  timeout 900 python bench.py --precision f16 --steps 20 --warmup 3 --no-extras --no-cpu-baseline $g > $O/f16_${i}_x$g.json 2> $O/f16_${i}_x$g.err; echo "fp16 family [$g] $(grep -o 'timed region.*' $O/f16_${i}_x$g.err)"
done; done

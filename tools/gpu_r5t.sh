#!/bin/bash
# Round 5: the exact family's non-temporal stores, done properly: product (EAGLE_F32_STORE_NT = 1, abl0) against the same source with plain stores (abl46), same graph mode, 20 steps after
# 3 warm-up steps (captures outside the timed region), alternating.
tag=${1:-r05t}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R/tools/convbench; mkdir -p libs/abl0; cp $R/eagle_amd/libeagle_hip.so libs/abl0/; cd $R
for i in 1 2 3; do for m in 0 46; do
  EAGLE_HIP_LIB=$R/tools/convbench/libs/abl$m/libeagle_hip.so timeout 900 python bench.py --precision f32 --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/f32_${m}_$i.json 2> $O/f32_${m}_$i.err; echo "exact family lib abl$m $(grep -o 'timed region.*' $O/f32_${m}_$i.err)"
  EAGLE_HIP_LIB=$R/tools/convbench/libs/abl$m/libeagle_hip.so timeout 900 python bench.py --height 1080 --width 1920 --detector l --imgsz 960 --batch 25 --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/cfg3_${m}_$i.json 2> $O/cfg3_${m}_$i.err; echo "cfg3 default lib abl$m $(grep -o 'timed region.*' $O/cfg3_${m}_$i.err)"
  EAGLE_HIP_LIB=$R/tools/convbench/libs/abl$m/libeagle_hip.so timeout 900 python bench.py --steps 20 --warmup 3 --no-extras --no-cpu-baseline > $O/def_${m}_$i.json 2> $O/def_${m}_$i.err; echo "default lib abl$m $(grep -o 'timed region.*' $O/def_${m}_$i.err)"
done; done

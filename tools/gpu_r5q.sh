#!/bin/bash
# Round 5: the fp16 family's stores non-temporal (libs/abl45) against the product, alternating.
tag=${1:-r05q}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R/tools/convbench; mkdir -p libs/abl0; cp $R/eagle_amd/libeagle_hip.so libs/abl0/; cd $R
for i in 1 2 3; do for m in 0 45; do
  EAGLE_HIP_LIB=$R/tools/convbench/libs/abl$m/libeagle_hip.so timeout 900 python bench.py --precision f16 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/f16_${m}_$i.json 2> $O/f16_${m}_$i.err; echo "fp16 family lib abl$m $(grep -o 'timed region.*' $O/f16_${m}_$i.err)"
done; done

#!/bin/bash
# Round 5: cache-policy bits on the halo LDS-DMA of the 32x32x16 kernel (libs/abl1 = sc0, abl16 = sc1, abl17 = sc0 + sc1; abl0 = product, abl8 = cache-hot halo for reference),
# isolated on 192->192 (v21) and 96->96 (v24); then the GPU suite with the new default (EAGLE_CONV_M32 default 6).   Usage: tools/gpu_r5h.sh <tag>
tag=${1:-r05h}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R/tools/convbench
mkdir -p libs/abl0; cp $R/eagle_amd/libeagle_hip.so libs/abl0/
LAYER=3,1,192,192,34,60,50 ONLY=21 ABLS="0 1 16 17 8" ./ablate_split.sh run 3 > $O/haux_192.txt 2>&1; cat $O/haux_192.txt
LAYER=3,1,96,96,68,120,50 ONLY=24 ABLS="0 1 16 17" ./ablate_split.sh run 3 > $O/haux_96.txt 2>&1; cat $O/haux_96.txt
cd $R
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log | tail -1; grep -E "^FAILED|^ERROR" $O/gpu_tests.log | head

#!/bin/bash
# Round 5: non-temporal hints — halo LDS-DMA (abl2 = nt, abl18 = sc1 + nt), output stores (abl32 = nt) — against the product (abl0), isolated on three layers.
tag=${1:-r05j}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R/tools/convbench
mkdir -p libs/abl0; cp $R/eagle_amd/libeagle_hip.so libs/abl0/
LAYER=3,1,192,192,34,60,50 ONLY=21 ABLS="0 2 18 32 8" ./ablate_split.sh run 3 > $O/nt_192.txt 2>&1; cat $O/nt_192.txt
LAYER=3,1,96,96,68,120,50 ONLY=24 ABLS="0 2 18 32" ./ablate_split.sh run 3 > $O/nt_96.txt 2>&1; cat $O/nt_96.txt

#!/bin/bash
# A-direct conv experiment on the GPU box: plain and TIMING builds, B = 50
cd $GRAFT_REPO_ROOT/tools/convbench
O=$GRAFT_REPO_ROOT/gpurun_out/ad; mkdir -p $O
{ for b in $ADBINS; do echo "== $b"; timeout 120 ./$b 50 1; done; } 2>&1 | tee $O/ad_$ADTAG.log

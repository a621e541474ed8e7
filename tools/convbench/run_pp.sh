#!/bin/bash
# A/B on the GPU box: tuned kernels vs the persistent pipelined variants, B = 50
cd $GRAFT_REPO_ROOT/tools/convbench
for rep in 1 2; do
  for c in 96 192 384; do
    echo "== cin $c (tuned)"; ONLY=$c ./bench_pp 50 | grep -v "\[check\]"
    for cfg in "32 3 8" "32 4 8" "32 6 8" "32 3 9" "32 6 9" "16 3 8" "16 6 8"; do set -- $cfg; KC=$1 NT=$2 VAR=$3 ONLY=$c ./bench_pp 50 | grep -v "\[check\]\|unsupported"; done
  done
done
ONLY=192 KC=32 NT=4 VAR=8 ./bench_pp 50 | grep check
for w in 1 2 3; do echo "== PP_WGS=$w"; EAGLE_CONV_PP_WGS=$w KC=32 NT=4 VAR=8 ONLY=192 ./bench_pp 50 | grep -v check; done

#!/bin/bash
# ping-pong conv experiment on the GPU box: plain and TIMING builds, B = 50
cd $GRAFT_REPO_ROOT/tools/convbench
O=$GRAFT_REPO_ROOT/gpurun_out/pp; mkdir -p $O
{ timeout 120 ./bench_pingpong 50 1; timeout 120 ./bench_pingpong_t 50 1; ONLY192=1 timeout 60 ./bench_g8_v0 50 1; } 2>&1 | tee $O/pp.log

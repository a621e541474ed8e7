#!/bin/bash
# GPU-box pass for the LK reproducibility investigation.  Usage: tools/gpu_lk_probe.sh <tag> [experiments...]
tag=${1:-lk}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 1500 python tools/probe_lk_concurrency.py "$@" > $O/probe.log 2>&1; grep -v "^\[" $O/probe.log | tail -60

#!/bin/bash
# GPU-box pass: GPU tests, LK co-run probe, bench (+cadence, +host frames).  Usage: tools/gpu_check.sh <tag> [bench args]
tag=${1:-chk}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/gpu_tests.log 2>&1; tail -3 $O/gpu_tests.log
PROBE_REPS=60 timeout 600 python tools/probe_lk_concurrency.py h_mfma f16v0/126 f16v0 f16nt4 f16 > $O/probe.log 2>&1; grep "non-repro\|still differ" $O/probe.log
timeout 900 python bench.py --cadence 25 --host-frames "$@" > $O/bench.json 2> $O/bench.err; cat $O/bench.json

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6f
timeout 600 python tools/probes/bneck_debug.py 8 512 256 2176 > gpurun_out/r6f/bneck_debug.log 2>&1; cat gpurun_out/r6f/bneck_debug.log

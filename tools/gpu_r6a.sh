cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6a
timeout 900 python -m pytest tests/test_gpu_bneck.py -x -q -m gpu > gpurun_out/r6a/bneck_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r6a/bneck_tests.log
tail -15 gpurun_out/r6a/bneck_tests.log
timeout 600 python tools/probes/bneck_probe.py 50 20 > gpurun_out/r6a/bneck_probe.log 2>&1; cat gpurun_out/r6a/bneck_probe.log

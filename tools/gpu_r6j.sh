cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6j
timeout 900 python -m pytest tests/test_gpu_bneck.py -x -q -m gpu > gpurun_out/r6j/bneck_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r6j/bneck_tests.log
tail -5 gpurun_out/r6j/bneck_tests.log
timeout 600 python tools/probes/bneck_debug.py 8 512 1024 > gpurun_out/r6j/bneck_debug.log 2>&1; grep "bad values" gpurun_out/r6j/bneck_debug.log
for f in 0 1; do EAGLE_BNECK_FORM=$f timeout 300 python tools/probes/bneck_probe.py 50 20 256 2>&1 | tail -1; done

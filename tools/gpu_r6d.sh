cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6d
EAGLE_BNECK_TIMING=1 EAGLE_HIP_LIB=$PWD/eagle_amd/libeagle_timing.so timeout 300 python tools/probes/bneck_probe.py 50 5 256,64 > gpurun_out/r6d/bneck_phase_timing.log 2>&1; cat gpurun_out/r6d/bneck_phase_timing.log
timeout 1500 python -m pytest tests/test_gpu_detectors.py -x -q -m gpu > gpurun_out/r6d/detector_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6d/detector_tests.log; tail -12 gpurun_out/r6d/detector_tests.log
timeout 2400 python -m pytest tests -x -q -m gpu --deselect tests/test_gpu_detectors.py > gpurun_out/r6d/gpu_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r6d/gpu_tests.log; tail -12 gpurun_out/r6d/gpu_tests.log

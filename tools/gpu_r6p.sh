cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6p
timeout 900 python -m pytest tests/test_gpu_bneck.py -x -q -m gpu > gpurun_out/r6p/bneck_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r6p/bneck_tests.log
tail -5 gpurun_out/r6p/bneck_tests.log
L=gpurun_out/r6p/bneck_v6.log
: > $L
for r in 1 0; do for f in 0 1; do echo "ring $r form $f:" >> $L; EAGLE_BNECK_RING=$r EAGLE_BNECK_FORM=$f timeout 300 python tools/probes/bneck_probe.py 50 20 256,64 >> $L 2>&1; done; done
for f in 0 1; do echo "B-direct, form $f timing build:" >> $L; EAGLE_BNECK_FORM=$f EAGLE_BNECK_TIMING=1 EAGLE_HIP_LIB=$PWD/eagle_amd/libeagle_timing.so timeout 300 python tools/probes/bneck_probe.py 50 5 256 >> $L 2>&1; done
cat $L

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6c
timeout 900 python -m pytest tests/test_gpu_bneck.py -x -q -m gpu > gpurun_out/r6c/bneck_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r6c/bneck_tests.log
tail -8 gpurun_out/r6c/bneck_tests.log
L=gpurun_out/r6c/bneck_ablation.log
echo "product:" > $L; timeout 300 python tools/probes/bneck_probe.py 50 20 256,64 >> $L 2>&1
for n in 1 2 3 4 5; do echo "ablation $n:" >> $L; EAGLE_HIP_LIB=$PWD/eagle_amd/libeagle_abl$n.so timeout 300 python tools/probes/bneck_probe.py 50 20 256 >> $L 2>&1; done
cat $L

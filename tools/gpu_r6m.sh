cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6m
L=gpurun_out/r6m/bneck_v3_phase_ablation.log
: > $L
for n in 1 2; do echo "timing build, ablation $n (1: no x traffic, 2: no MFMA):" >> $L; EAGLE_BNECK_TIMING=1 EAGLE_HIP_LIB=$PWD/eagle_amd/libeagle_t$n.so timeout 300 python tools/probes/bneck_probe.py 50 5 256 >> $L 2>&1; done
cat $L

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6g
L=gpurun_out/r6g/bneck_debug.log
echo "product" > $L; timeout 600 python tools/probes/bneck_debug.py 8 512 >> $L 2>&1
for n in 6 7 8; do echo "ablation $n" >> $L; EAGLE_HIP_LIB=$PWD/eagle_amd/libeagle_abl$n.so timeout 600 python tools/probes/bneck_debug.py 8 512 >> $L 2>&1; done
cat $L

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6b
L=gpurun_out/r6b/bneck_ablation.log
echo "product:" > $L; timeout 300 python tools/probes/bneck_probe.py 50 20 256 >> $L 2>&1
for n in 1 2 3 4 5; do echo "ablation $n:" >> $L; EAGLE_HIP_LIB=$PWD/eagle_amd/libeagle_abl$n.so timeout 300 python tools/probes/bneck_probe.py 50 20 256 >> $L 2>&1; done
for w in 128 512 1024 6800; do echo "EAGLE_BNECK_WGS=$w:" >> $L; EAGLE_BNECK_WGS=$w timeout 300 python tools/probes/bneck_probe.py 50 20 256 >> $L 2>&1; done
cat $L

cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6h
L=gpurun_out/r6h/bneck_debug.log
echo "product, LDS pad 30000 (one workgroup per CU)" > $L; EAGLE_BNECK_LDS_PAD=30000 timeout 600 python tools/probes/bneck_debug.py 8 512 >> $L 2>&1
for n in 3 5 2; do echo "ablation $n" >> $L; EAGLE_HIP_LIB=$PWD/eagle_amd/libeagle_abl$n.so timeout 600 python tools/probes/bneck_debug.py 8 512 >> $L 2>&1; done
grep -n "^product\|^ablation\|bad values\|form0" $L | head -60

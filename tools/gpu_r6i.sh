cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r6i
L=gpurun_out/r6i/bneck_debug.log
: > $L
for pad in 512 1024 4096 8192; do echo "product, LDS pad $pad" >> $L; EAGLE_BNECK_LDS_PAD=$pad timeout 600 python tools/probes/bneck_debug.py 8 512 >> $L 2>&1; done
grep -n "^product\|bad values" $L | head -60

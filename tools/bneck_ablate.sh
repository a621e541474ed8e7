#!/bin/bash
# Developer ablation of the fused Bottleneck kernel (csrc/bneck.hip, -DEAGLE_ABL_BNECK=n): builds one library per ablation next to the product library and times the
# isolated launch at B = 50 with tools/probes/bneck_probe.py.  usage: bash tools/bneck_ablate.sh "0 1 2 3 4 5" [extra hipcc flags]
set -e
cd "$(dirname "$0")/../eagle_amd/csrc"
OBJS=$(ls *.o | grep -v '^bneck' | tr '\n' ' ')
for n in $1; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -ffp-contract=off -DEAGLE_ABL_BNECK=$n $2 -c bneck.hip -o /tmp/bneck_abl$n.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libeagle_abl$n.so $OBJS /tmp/bneck_abl$n.o -ldl
done

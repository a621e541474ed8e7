#!/bin/bash
# Ablation of the split family's generic kernel on the 48->48 @135x240 layer: builds the library with -DEAGLE_ABL=n
# (conv_kernels.inc: 1 no MFMAs, 2 no activation loads after chunk 0, 3 no output stores, 4 all three, 5 / 6 also no weight loads / no LDS staging
# writes after chunk 0, 7 only the memory traffic removed, 8 also no LDS fragment reads, 9 also no epilogue) into tools/convbench/libs/abl<n>/.
# `ablate_split.sh build` in the build container — in a COPY of the source tree, so that the product's objects are never touched (an earlier form of this
# script rebuilt in place and left ablated objects in the product library; its saturation report caught it) — `ablate_split.sh run [reps]` on the GPU box
# (alternating over the builds).  ABLS="0 6 8" restricts the set.  The A-direct kernel: SYM=EAGLE_ABL_AD ABLS="0 1 2 3 4 5 6 7" ... build, then
# LAYER=3,1,192,192,34,60,50 ONLY=8 ABLS=... run.
cd "$(dirname "$0")"
ABLS=${ABLS:-0 1 2 3 4 5 6 7 8 9}
if [ "$1" = build ]; then
  T=$(mktemp -d); mkdir -p $T/eagle_amd $T/include
  cp -r ../../eagle_amd/csrc $T/eagle_amd/; cp ../../eagle_amd/pitch.py $T/eagle_amd/; cp ../../include/eagle.h $T/include/
  for n in $ABLS; do
    mkdir -p libs/abl$n
    make -C $T/eagle_amd/csrc clean > /dev/null
    make -C $T/eagle_amd/csrc -j8 EXTRA=-D${SYM:-EAGLE_ABL}=$n > /dev/null || exit 1
    cp $T/eagle_amd/libeagle_hip.so libs/abl$n/
  done
  rm -rf $T
  exit 0
fi
R=${2:-3}
printf "${LAYER:-3,1,48,48,135,240,50}\n" > /tmp/l48.csv
for r in $(seq $R); do for n in $ABLS; do
  LD_LIBRARY_PATH=libs/abl$n:$LD_LIBRARY_PATH TUNE_ONLY=${ONLY:-18} ./split_tune.out /tmp/l48.csv 2>/dev/null | awk -F, -v n=$n '{print "abl" n, "res" $13, $14}'
done; done | awk '{k=$1" "$2; s[k]+=$3; c[k]++; if(!(k in m)||$3<m[k]) m[k]=$3} END {for (k in s) printf "%s mean %.1f min %.1f us\n", k, s[k]/c[k], m[k]}' | sort

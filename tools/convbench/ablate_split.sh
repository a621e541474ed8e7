#!/bin/bash
# Ablation of the split family's generic kernel on the 48->48 @135x240 layer: builds the library with -DEAGLE_ABL=0..7
# (conv_kernels.inc: 1 no MFMAs, 2 no activation loads after chunk 0, 3 no output stores, 4 all three, 5 / 6 also no weight loads / no LDS staging writes after chunk 0, 7 only the memory traffic removed) into tools/convbench/libs/abl<n>/ — run
# `ablate_split.sh build` in the build container, `ablate_split.sh run [reps]` on the GPU box (alternating over the builds).
cd "$(dirname "$0")"
if [ "$1" = build ]; then
  for n in ${ABLS:-0 1 2 3 4 5 6 7 8 9}; do
    mkdir -p libs/abl$n
    touch ../../eagle_amd/csrc/conv_inst_s0.hip
    make -C ../../eagle_amd/csrc -j8 EXTRA=-DEAGLE_ABL=$n > /dev/null || exit 1
    cp ../../eagle_amd/libeagle_hip.so libs/abl$n/
  done
  touch ../../eagle_amd/csrc/conv_inst_s0.hip; make -C ../../eagle_amd/csrc -j8 > /dev/null || { echo 'PRODUCT BUILD FAILED'; exit 1; }      # back to the product build
  exit 0
fi
R=${2:-3}
printf "3,1,48,48,135,240,50\n" > /tmp/l48.csv
for r in $(seq $R); do for n in ${ABLS:-0 1 2 3 4 5 6 7 8 9}; do
  LD_LIBRARY_PATH=libs/abl$n:$LD_LIBRARY_PATH TUNE_ONLY=18 ./split_tune.out /tmp/l48.csv 2>/dev/null | awk -F, -v n=$n '{print "abl" n, "res" $13, $14}'
done; done | awk '{k=$1" "$2; s[k]+=$3; c[k]++; if(!(k in m)||$3<m[k]) m[k]=$3} END {for (k in s) printf "%s mean %.1f min %.1f us\n", k, s[k]/c[k], m[k]}' | sort

#!/bin/bash
# usage: ab_lib.sh <layers.csv> <TUNE_ONLY spec per layer or ""> <libdirA> <libdirB> [reps]: alternates split_tune.out against two builds of the
# library (tools/convbench/libs/<name>/libeagle_hip.so) on the same box and prints the best configuration's time per layer and build.
L=$1; ONLY=$2; A=$3; B=$4; R=${5:-3}
cd "$(dirname "$0")"
for r in $(seq $R); do for v in $A $B; do
  if [ -n "$ONLY" ]; then export TUNE_ONLY=$ONLY; else unset TUNE_ONLY; fi
  LD_LIBRARY_PATH=libs/$v:$LD_LIBRARY_PATH ./split_tune.out $L 2>/dev/null | awk -F, -v v=$v '{k=$2","$3","$4","$5","$6","$13; if(!(k in m)||$14<m[k]) {m[k]=$14; c[k]=$9","$10","$11","$12}} END {for (k in m) print v, k, m[k], c[k]}'
done; done | awk '{k=$1" "$2; s[k]+=$3; n[k]++; if(!(k in m)||$3<m[k]) m[k]=$3; c[k]=$4} END {for (k in s) printf "%s mean %.1f min %.1f cfg %s\n", k, s[k]/n[k], m[k], c[k]}' | sort -k2,2 -k1,1

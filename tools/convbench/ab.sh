#!/bin/bash
# usage: ab.sh <variantA> <variantB> [reps]   -- alternates ./bench_<variant> on the three 3x3 classes and prints mean/min us
A=$1; B=$2; R=${3:-4}
for r in $(seq $R); do for v in $A $B; do for c in 96 192 384; do
  us=$(ONLY=$c ./bench_$v 50 | grep -v "\[check\]" | sed -E 's/.* ([0-9.]+) us .*/\1/')
  echo "$v $c $us"
done; done; done | awk '{k=$1" "$2; s[k]+=$3; n[k]++; if(!(k in m)||$3<m[k]) m[k]=$3} END {for (k in s) printf "%s mean %.1f min %.1f\n", k, s[k]/n[k], m[k]}' | sort

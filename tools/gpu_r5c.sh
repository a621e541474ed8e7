#!/bin/bash
# Round 5, third GPU pass: (1) the multirank composition again, four times, with exit codes (r05a: died silently after the timed region, r05b: fine);
# (2) ablation of the 32x32x16 A-direct kernel (tools/convbench/libs/abl<n> built with -DEAGLE_ABL_M32=n) next to the 16x16 form's table of round 4;
# (3) weight ring five steps ahead (EAGLE_CONV_M32_RING=6); (4) latency modes at B = 12, 16, 25 (where does small-batch mode stop paying);
# (5) the new tests; (6) SQ counters with and without EAGLE_CONV_M32.   Usage: tools/gpu_r5c.sh <tag>
tag=${1:-r05c}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
for i in 1 2 3 4; do
  if [ $((i % 2)) = 0 ]; then export PYTHONFAULTHANDLER=; else unset PYTHONFAULTHANDLER; fi
  timeout 600 python bench.py --force-multirank-path --backend nccl --gather rccl --steps 20 --warmup 5 --no-cpu-baseline > $O/mr_$i.json 2> $O/mr_$i.err; rc=$?
  echo "multirank run $i rc=$rc json_bytes=$(stat -c %s $O/mr_$i.json) $(grep -o 'timed region.*' $O/mr_$i.err)"; [ $rc != 0 ] && tail -25 $O/mr_$i.err
done
timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/default_1.json 2> $O/default_1.err; grep -o 'timed region.*' $O/default_1.err
timeout 900 python -m pytest tests/test_gpu_pipeline.py tests/test_gpu_edges.py -m gpu -q -x -k "small_batch or batch_and_position or strided or multirank_bench_uses or graph" > $O/new_tests.log 2>&1; tail -3 $O/new_tests.log
cd $R/tools/convbench
cp $R/eagle_amd/libeagle_hip.so libs/abl0/ 2>/dev/null || { mkdir -p libs/abl0; cp $R/eagle_amd/libeagle_hip.so libs/abl0/; }
LAYER=3,1,192,192,34,60,50 ONLY=21 ABLS="0 1 2 3 4 5 6 7" ./ablate_split.sh run 3 > $O/m32_ablation_192.txt 2>&1; cat $O/m32_ablation_192.txt
LAYER=3,1,96,96,68,120,50 ONLY=22 ABLS="0 1 2 3 4 5 6 7" ./ablate_split.sh run 3 > $O/m32_ablation_96.txt 2>&1; cat $O/m32_ablation_96.txt
for r in 1 2 3; do for ring in 3 6; do for v in 21 22; do
  EAGLE_CONV_M32_RING=$ring TUNE_ONLY=$v ./split_tune.out layers_m32.csv 2>/dev/null | awk -F, -v ring=$ring '$13==0 {print ring, $0}'
done; done; done | awk '{split($2,a,","); k=a[4]"->"a[5]" v"a[12]" ring"$1; s[k]+=a[14]; c[k]++} END {for (k in s) printf "%s mean %.1f us\n", k, s[k]/c[k]}' | sort > $O/m32_ring.txt; cat $O/m32_ring.txt
cd $R
LATENCY_BATCHES=12,16,25 timeout 900 python bench.py --latency-only --latency-calls 60 > $O/latency_modes_big.json 2> $O/latency_modes_big.err; grep "latency B" $O/latency_modes_big.err
EAGLE_CONV_M32=0 bash tools/pmc_sq.sh $tag/sq_m32_0 > $O/sq_m32_0.txt 2>&1; grep -E "conv_split_ad" $O/sq_m32_0.txt
EAGLE_CONV_M32=1 bash tools/pmc_sq.sh $tag/sq_m32_1 > $O/sq_m32_1.txt 2>&1; grep -E "conv_split_ad" $O/sq_m32_1.txt

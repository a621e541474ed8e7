#!/bin/bash
# Round 5, fourth GPU pass: (1) the Winograd F(2x2, 3x3) harness (tools/convbench/wino_main.hip: correctness on two small shapes, then B = 50 timing on
# 192->192 @34x60 and 384->384 @17x30) alternating with the direct A-direct forms (variants 8 / 21) of the same layers; (2) the two process compositions
# again, now with the timed region's parts logged and eagle_gather's staging cached; (3) the whole GPU suite (small-batch mode is the default for batch <= 8 now);
# (4) the default bench line.   Usage: tools/gpu_r5d.sh <tag>
tag=${1:-r05d}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R/tools/convbench
printf "3,1,192,192,34,60,50\n3,1,384,384,17,30,50\n" > /tmp/lw.csv
for r in 1 2 3; do
  ./wino.out 6 | sed "s/^/rep$r /"
  ./wino_r2.out 6 | sed "s/^/rep$r /"
  for v in 8 21; do TUNE_ONLY=$v ./split_tune.out /tmp/lw.csv 2>/dev/null | awk -F, -v r=$r '$13==0 {print "rep" r, "DIRECT v" $12, $4 "->" $5 "@" $6 "x" $7, $14}'; done
done > $O/wino_vs_direct.txt 2>&1; cat $O/wino_vs_direct.txt
cd $R
for i in 1 2; do
  timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $O/ab_default_$i.json 2> $O/ab_default_$i.err; grep -o 'timed region.*' $O/ab_default_$i.err
  timeout 600 python bench.py --force-multirank-path --backend nccl --gather rccl --steps 20 --warmup 5 --no-cpu-baseline > $O/ab_multirank_$i.json 2> $O/ab_multirank_$i.err; grep -o 'timed region.*' $O/ab_multirank_$i.err; wc -l $O/ab_multirank_$i.json
done
timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log | tail -1; grep -E "^FAILED|^ERROR" $O/gpu_tests.log | head
timeout 1500 python bench.py > $O/bench.json 2> $O/bench.err; wc -l $O/bench.json; python3 -c "
import json; d=json.loads(open('$O/bench.json').readline()); print(d['value'], d['roofline']['frac'], d['exact_family']['value'], d['cfg3']['default']['value']); [print(r) for r in d['latency']['rows']]"

#!/bin/bash
# One GPU-box pass of round 5: tests, default bench, the multi-rank composition at world 1, kernel-trace profiles + PMC traffic passes of the same command, the corrected ablation 8.
# Usage: tools/gpu_round5.sh <tag> [skip_tests]
tag=${1:-x}
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
if [ -z "$2" ]; then timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log | tail -1; fi
timeout 1800 python bench.py --cadence 25 > $O/bench.json 2> $O/bench.err; wc -l $O/bench.json; tail -c 300 $O/bench.json; grep -v parity $O/bench.err | tail -4
timeout 600 python bench.py --force-multirank-path --backend nccl --gather rccl --no-cpu-baseline > $O/bench_multirank_world1.json 2> $O/bench_multirank_world1.err; wc -l $O/bench_multirank_world1.json; grep -o 'timed region.*' $O/bench_multirank_world1.err
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extras > $O/prof.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-extras > $O/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-extras > $O/pmc_write.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32 -- python3 $R/bench.py --precision f32 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_f32.log 2>&1
cd $R && python3 -c "import __graft_entry__ as g; g.smoke(); print(\"smoke ok\")" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
cd $R && python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/conv_hbm_traffic_f32s.json --steps 2 --warmup 0 --no-cpu-baseline --precision f32s
bash tools/pmc_sq.sh $tag/sq > $O/sq.txt 2>&1; head -8 $O/sq.txt
cd $R/tools/convbench; mkdir -p libs/abl0; cp $R/eagle_amd/libeagle_hip.so libs/abl0/
LAYER=3,1,192,192,34,60,50 ONLY=21 ABLS="0 3 8" ./ablate_split.sh run 3 > $O/m32_ablation8_corrected.txt 2>&1; cat $O/m32_ablation8_corrected.txt
cd $R
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -size +20M -delete
ls -R $O | head -50

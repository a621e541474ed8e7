#!/bin/bash
# One parameterised GPU-box script (round 6 housekeeping: replaces the one-shot tools/gpu_r5*.sh / gpu_r6*.sh files; what each historical run did is listed in
# tools/gpu_runs.md).  usage (through gpurun):  bash tools/gpu_run.sh <tag> <step> [<step> ...]      outputs under gpurun_out/<tag>/
#   tests            the whole -m gpu suite
#   tests:<expr>     pytest -k <expr> of the -m gpu suite
#   smoke            __graft_entry__.smoke()
#   bench            default bench line (all extras)                      -> bench_default_1gpu.json / .err
#   bench:<args>     bench.py with the given arguments (quote them)       -> bench_custom.json
#   multirank        the WORLD_SIZE > 1 process composition at world 1 (nccl group + library RCCL)
#   selflaunch       `python bench.py --gpus 2 --shared-gpu --backend gloo` without a launcher
#   prof             rocprofv3 --kernel-trace --stats of the default step (f32s)         -> prof/ (kernel_stats csv kept)
#   prof_f32         the same for --precision f32
#   pmc              rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE passes (separate runs) + tools/pmc_traffic.py -> conv_hbm_traffic_f32s.json
#   sq               SQ stall counters (tools/pmc_sq.sh)
#   bneck            fused-Bottleneck probe: both tile forms, both phase-1 forms, Cin 256 / 64 (tools/probes/bneck_probe.py)
#   bneck_ab         whole-pipeline A/B EAGLE_BNECK_FUSED=0/1, three alternating pairs
#   bneck_ds_ab      whole-pipeline A/B EAGLE_BNECK_DS=0/1 (block 0's downsample branch inside the fused launch), three alternating pairs
#   bneck_ds         operator-level and pipeline probes of the downsample-fused launch (tools/probes/bneck_ds_probe.py, bneck_ds_pipeline_probe.py)
#   env_ab:<VAR=v>   whole-pipeline A/B of one environment knob (unset / set), three alternating pairs, no extras          -> env_ab.log
#   lat_ab:<VAR=v>   one-, two- and four-frame call latency with the knob unset / set, two alternating pairs (tools/probes/latency_env_ab.sh)
#   b1trace          rocprofv3 --kernel-trace of 40 one-frame calls + tools/probes/latency_b1_trace.py parse                -> b1_critical_path.txt
#   cpu_sweep        cpu_baseline thread sweep (16 32 64 128)
tag=${1:-x}; shift
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/$tag; mkdir -p $O
cd $R
for step in "$@"; do
  case "$step" in
    tests) timeout 2400 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; grep -E "passed|failed" $O/gpu_tests.log | tail -1 ;;
    tests:*) timeout 2400 python -m pytest tests -m gpu -q -k "${step#tests:}" > $O/gpu_tests_k.log 2>&1; grep -E "passed|failed" $O/gpu_tests_k.log | tail -1 ;;
    smoke) python3 -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > $O/smoke.log 2>&1; tail -1 $O/smoke.log ;;
    bench) timeout 1800 python bench.py > $O/bench_default_1gpu.json 2> $O/bench_default_1gpu.err; wc -c $O/bench_default_1gpu.json ;;
    bench:*) timeout 1800 python bench.py ${step#bench:} > $O/bench_custom.json 2> $O/bench_custom.err; tail -c 400 $O/bench_custom.json ;;
    multirank) timeout 900 python bench.py --force-multirank-path --backend nccl --gather rccl --no-cpu-baseline --no-extras > $O/bench_multirank_world1.json 2> $O/bench_multirank_world1.err; tail -c 300 $O/bench_multirank_world1.json ;;
    selflaunch) timeout 900 python bench.py --gpus 2 --shared-gpu --backend gloo --no-cpu-baseline --no-extras --steps 4 --warmup 1 --batch 8 > $O/bench_selflaunch_2ranks_1gpu.json 2> $O/bench_selflaunch_2ranks_1gpu.err; tail -c 300 $O/bench_selflaunch_2ranks_1gpu.json ;;
    prof) (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python3 $R/bench.py --steps 6 --warmup 1 --no-cpu-baseline --no-extras > $O/prof.log 2>&1); find $O/prof -name "*kernel_trace.csv" -delete ;;
    prof_f32) (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f32 -- python3 $R/bench.py --precision f32 --steps 4 --warmup 1 --no-cpu-baseline --no-extras > $O/prof_f32.log 2>&1); find $O/prof_f32 -name "*kernel_trace.csv" -delete ;;
    pmc) (cd /tmp && export TMPDIR=/tmp && rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-extras > $O/pmc_fetch.log 2>&1
          rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 $R/bench.py --steps 2 --warmup 0 --no-cpu-baseline --no-extras > $O/pmc_write.log 2>&1)
         python3 tools/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/conv_hbm_traffic_f32s.json --steps 2 --warmup 0 --no-cpu-baseline --precision f32s
         find $O -name "*counter_collection.csv" -size +20M -delete ;;
    sq) bash tools/pmc_sq.sh $tag/sq > $O/sq.txt 2>&1; head -8 $O/sq.txt ;;
    bneck) L=$O/bneck_probe.log; : > $L
           for p1 in ring direct; do for f in 0 1; do echo "phase 1 $p1, form $f:" >> $L; EAGLE_BNECK_P1=$p1 EAGLE_BNECK_FORM=$f timeout 300 python tools/probes/bneck_probe.py 50 20 256,64 >> $L 2>&1; done; done; cat $L ;;
    bneck_ab) L=$O/bneck_pipeline_ab.log; : > $L
           for rep in 1 2 3; do for f in 0 1; do echo "EAGLE_BNECK_FUSED=$f rep $rep" >> $L
             EAGLE_BNECK_FUSED=$f timeout 600 python bench.py --no-extras --no-cpu-baseline --latency-calls 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'frames/s', d['ms_per_step'], 'ms/step; conv family', d['roofline']['conv_ms_per_step'], 'ms, frac', d['roofline']['frac'])" >> $L 2>&1
           done; done; cat $L ;;
    bneck_ds_ab) L=$O/bneck_ds_pipeline_ab.log; : > $L
           for rep in 1 2 3; do for f in 0 1; do echo "EAGLE_BNECK_DS=$f rep $rep" >> $L
             EAGLE_BNECK_DS=$f timeout 600 python bench.py --no-extras --no-cpu-baseline --latency-calls 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'frames/s', d['ms_per_step'], 'ms/step; conv family', d['roofline']['conv_ms_per_step'], 'ms, frac', d['roofline']['frac'])" >> $L 2>&1
           done; done; cat $L ;;
    bneck_ds) timeout 600 python tools/probes/bneck_ds_probe.py > $O/bneck_ds_probe.log 2>&1; timeout 600 python tools/probes/bneck_ds_pipeline_probe.py > $O/bneck_ds_pipeline_probe.log 2>&1; tail -4 $O/bneck_ds_probe.log $O/bneck_ds_pipeline_probe.log ;;
    env_ab:*) L=$O/env_ab.log; kv=${step#env_ab:}
           for rep in 1 2 3; do for f in 0 1; do if [ $f = 0 ]; then echo "unset rep $rep" >> $L; pre=""; else echo "$kv rep $rep" >> $L; pre="$kv"; fi
             env $pre timeout 600 python bench.py --no-extras --no-cpu-baseline --latency-calls 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['value'], 'frames/s', d['ms_per_step'], 'ms/step; conv family', d['roofline']['conv_ms_per_step'], 'ms, frac', d['roofline']['frac'])" >> $L 2>&1
           done; done; cat $L ;;
    lat_ab:*) bash tools/probes/latency_env_ab.sh $tag ${step#lat_ab:} ;;
    b1trace) (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $O/b1trace -- python3 $R/tools/probes/latency_b1_trace.py run > $O/b1trace.log 2>&1)
             python3 tools/probes/latency_b1_trace.py parse $O/b1trace > $O/b1_critical_path.txt 2>&1; find $O/b1trace -name "*kernel_trace.csv" -delete; head -30 $O/b1_critical_path.txt ;;
    cpu_sweep) for t in 16 32 64 128; do echo "threads $t" >> $O/cpu_baseline_thread_sweep.txt; timeout 600 python bench.py --cpu-baseline-only --cpu-threads $t --cpu-frames 40 2>/dev/null | tail -1 >> $O/cpu_baseline_thread_sweep.txt; done; cat $O/cpu_baseline_thread_sweep.txt ;;
    *) echo "unknown step $step" ;;
  esac
done
ls $O | head -40

#!/bin/bash
# same-box A/B of two builds of the library: tools/convbench/libA.so vs libB.so (bench.py, no CPU baseline), alternating
for rep in 1 2; do
  for v in ${VARIANTS:-A B}; do
    cp tools/convbench/lib$v.so eagle_amd/libeagle_hip.so
    python bench.py --no-cpu-baseline 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v', j['value'], 'fps  conv', j['roofline']['achieved'], 'TFLOP/s', j['roofline']['conv_ms_per_step'], 'ms/step')"
  done
done

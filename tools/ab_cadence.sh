#!/bin/bash
# same-box A/B of the reference-cadence number for two library builds (tools/convbench/libA.so, libB.so)
for rep in 1 2; do
  for v in ${VARIANTS:-A B}; do
    cp tools/convbench/lib$v.so eagle_amd/libeagle_hip.so
    python bench.py --no-cpu-baseline --cadence 25 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); print('$v', j['value'], 'fps   cadence', j['reference_cadence']['value'])"
  done
done

#!/bin/bash
# Samples board power and shader clock (rocm-smi, read-only) while the default bench's timed region runs: is the chip at its power limit under these kernels?
# usage: tools/probes/power_probe.sh [bench args]   -> prints the samples taken while the GPU was busy and their summary
cd "$(dirname "$0")/../.."
( for i in $(seq 200); do rocm-smi --showpower --showclocks --showmaxpower --json 2>/dev/null | tr -d '\n'; echo; sleep 0.1; done ) > /tmp/power_samples.jsonl &
SP=$!
python bench.py --no-cpu-baseline --no-extras --steps 60 --warmup 3 "$@" > /tmp/power_bench.json 2>/dev/null
kill $SP 2>/dev/null; wait $SP 2>/dev/null
python3 - <<'PY'
import json
rows=[]
for ln in open('/tmp/power_samples.jsonl'):
    ln=ln.strip()
    if not ln.startswith('{'): continue
    try: d=json.loads(ln)
    except Exception: continue
    c=d.get('card0',{})
    p=[v for k,v in c.items() if 'Power' in k and 'Max' not in k and 'W' in k]
    mx=[v for k,v in c.items() if 'Max' in k and 'Power' in k]
    sclk=[v for k,v in c.items() if 'sclk' in k.lower()]
    rows.append((p[0] if p else None, mx[0] if mx else None, sclk[0] if sclk else None))
print("samples", len(rows))
busy=[r for r in rows if r[0] and float(str(r[0]).split()[0])>300]
for r in rows[::10]: print(r)
if busy:
    w=[float(str(r[0]).split()[0]) for r in busy]
    print("busy samples", len(busy), "power mean %.0f W max %.0f W" % (sum(w)/len(w), max(w)), "cap", busy[0][1], "sclk e.g.", busy[len(busy)//2][2])
print(open('/tmp/power_bench.json').read()[:200])
PY

"""Developer probe: idle time between consecutive kernels of each stream in a rocprofv3 --kernel-trace of bench.py (how much of a step is launch gap rather than kernel).

    python3 tools/probes/stream_gaps.py <dir with *kernel_trace.csv> [frames_per_step]
"""
import csv, glob, os, sys
import numpy as np

d = sys.argv[1]
f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
rows = list(csv.DictReader(open(f)))
streams = {}
for r in rows:
    streams.setdefault((r["Queue_Id"], r.get("Stream_Id", "")), []).append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
t_all0 = min(int(r["Start_Timestamp"]) for r in rows); t_all1 = max(int(r["End_Timestamp"]) for r in rows)
print(f"{len(rows)} kernels, {len(streams)} (queue, stream) pairs, trace span {(t_all1 - t_all0) / 1e6:.1f} ms")
for key, ev in sorted(streams.items(), key=lambda kv: -len(kv[1])):
    ev.sort()
    if len(ev) < 50:
        continue
    # steady state: the last 60 % of the stream's kernels
    ev = ev[int(len(ev) * 0.4):]
    dur = np.array([e - s for s, e, _ in ev]) / 1e3
    gap = np.array([max(0, ev[i + 1][0] - ev[i][1]) for i in range(len(ev) - 1)]) / 1e3
    small = gap[gap < 100]                                   # gaps above 100 us are step boundaries / host waits, not launch gaps
    span = (ev[-1][1] - ev[0][0]) / 1e3
    names = {}
    for s, e, n in ev:
        k = n.split("(")[0].split("<")[0].replace("eagle::", "").replace("void ", ""); names[k] = names.get(k, 0) + 1
    top = ", ".join(f"{k} x{v}" for k, v in sorted(names.items(), key=lambda kv: -kv[1])[:3])
    print(f"queue {key[0]} stream {key[1]}: {len(ev)} kernels over {span / 1e3:.1f} ms: kernel time {dur.sum() / 1e3:.1f} ms ({dur.sum() / span:.3f} of the span), "
          f"gaps < 100 us: {len(small)} totalling {small.sum() / 1e3:.2f} ms (median {np.median(small):.1f} us, p90 {np.percentile(small, 90):.1f} us), larger gaps {gap[gap >= 100].sum() / 1e3:.1f} ms   [{top}]")
    if "--large" in sys.argv:
        for i in range(len(ev) - 1):
            g = (ev[i + 1][0] - ev[i][1]) / 1e3
            if g >= 100:
                print(f"    gap {g:9.1f} us after {ev[i][2][:60]} ({(ev[i][1] - ev[i][0]) / 1e3:.0f} us) before {ev[i + 1][2][:60]}")

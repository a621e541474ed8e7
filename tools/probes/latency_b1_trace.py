"""Developer probe: where a one-frame call spends its time.

    (cd /tmp && rocprofv3 --kernel-trace --output-format csv -d $OUT -- python3 $REPO/tools/probes/latency_b1_trace.py run)     # 40 calls of one frame, default handle
    python3 tools/probes/latency_b1_trace.py parse $OUT                                                                     # the last 20 calls: span, busy time, top kernels

`parse` groups the kernel trace into calls by the gaps between them (the host-side copy and record read-back leave > 150 us without a kernel), and reports per call: the
span from the first kernel's start to the last kernel's end, the union of the kernels' intervals (GPU busy), the longest kernels, and the time per kernel family.
"""
import csv, glob, os, sys
import numpy as np


def run():
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    import time
    from eagle_amd import lib, synth, weights
    h = lib.Handle(batch=1, **({"use_graph": 0} if "nograph" in sys.argv else {}))
    weights.load_into(h, [weights.make_hrnet_state_dict(0), weights.make_yolo_state_dict("n", 0)])
    clip = synth.clip(seed=0, n=4)
    out = np.zeros(1, lib.RESULT_DTYPE)
    ts = []
    for k in range(40):
        t0 = time.perf_counter()
        h.process(clip[k % 4:k % 4 + 1], out)
        ts.append((time.perf_counter() - t0) * 1e3)
        time.sleep(0.002)
    print("host-side ms per call (last 20): median %.3f  min %.3f" % (np.median(ts[20:]), np.min(ts[20:])))


def parse(d):
    f = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))[-1]
    rows = list(csv.DictReader(open(f)))
    ev = sorted(((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows))
    calls, cur = [], [ev[0]]
    for e in ev[1:]:
        if e[0] - max(x[1] for x in cur[-8:]) > 150_000:
            calls.append(cur); cur = []
        cur.append(e)
    calls.append(cur)
    calls = [c for c in calls if len(c) > 100][-20:]
    spans, busys = [], []
    fam = {}
    for c in calls:
        t0, t1 = c[0][0], max(x[1] for x in c)
        spans.append((t1 - t0) / 1e3)
        busy, end = 0, t0
        for s, e, _ in c:
            if e > end:
                busy += e - max(s, end); end = e
        busys.append(busy / 1e3)
        for s, e, name in c:
            key = name.split("(")[0].split("<")[0].replace("eagle::", "").replace("void ", "")
            fam.setdefault(key, [0.0, 0])
            fam[key][0] += (e - s) / 1e3 / len(calls); fam[key][1] += 1
    print(f"{len(calls)} calls, {len(calls[-1])} kernels per call")
    print("span first kernel start -> last kernel end: median %.1f us (min %.1f); union of kernel intervals: median %.1f us" % (np.median(spans), np.min(spans), np.median(busys)))
    print("per kernel family, sum of durations per call (concurrent streams overlap, so the sum exceeds the span):")
    for k, (us, cnt) in sorted(fam.items(), key=lambda kv: -kv[1][0])[:16]:
        print(f"  {k[:70]:70s} {us:9.1f} us  x{cnt // len(calls)}")
    c = calls[-1]
    t0 = c[0][0]
    print("last call, the 12 longest kernels (start offset, duration):")
    for s, e, name in sorted(c, key=lambda x: x[0] - x[1])[:12]:
        print(f"  +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:8.1f} us  {name[:90]}")
    print("last call, tail (the last 8 kernels):")
    for s, e, name in c[-8:]:
        print(f"  +{(s - t0) / 1e3:8.1f} us  {(e - s) / 1e3:8.1f} us  {name[:90]}")


if __name__ == "__main__":
    run() if sys.argv[1] == "run" else parse(sys.argv[2])
    if sys.argv[1] == "parse" and "--timeline" in sys.argv:      # the last call's kernels between two offsets (us): --timeline A B
        a0, a1 = float(sys.argv[sys.argv.index("--timeline") + 1]), float(sys.argv[sys.argv.index("--timeline") + 2])
        f = sorted(glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True))[-1]
        ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]) for r in csv.DictReader(open(f)))
        calls, cur = [], [ev[0]]
        for e in ev[1:]:
            if e[0] - max(x[1] for x in cur[-8:]) > 150_000:
                calls.append(cur); cur = []
            cur.append(e)
        calls.append(cur)
        c = [c for c in calls if len(c) > 100][-1]
        for s_, e_, n_, q_ in c:
            a = (s_ - c[0][0]) / 1e3
            if a0 <= a <= a1:
                print(f"{a:8.1f} {(e_ - s_) / 1e3:7.1f}  q{q_}  {n_.replace('void eagle::', '').replace('eagle::', '').split('(')[0][:50]}")

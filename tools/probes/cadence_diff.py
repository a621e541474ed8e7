"""Developer probe: where do the records of the 1000-frame cadence differ between device batches / runs?"""
import sys
import numpy as np
sys.path.insert(0, ".")
from eagle_amd import synth, weights
from eagle_amd.coordinate_model import CoordinateModel
hs, ys = weights.make_hrnet_state_dict(0), weights.make_yolo_state_dict("n", 0)
base = np.stack([synth.frame(0, t) for t in range(40)])
clip = np.ascontiguousarray(np.concatenate([base, base[::-1]] * 13)[:1000])
out = {}
for batch in (50, 8):
    m = CoordinateModel(precision="f16", batch=batch, hrnet_state_dict=hs, detector_state_dict=ys)
    out[batch] = m.flow_records(clip, 8, 25).copy()
    if batch == 50:
        out["again"] = m.flow_records(clip, 8, 25).copy()
    m.handle.close()
def diff(a, b, tag):
    bad = [i for i in range(len(a)) if a[i].tobytes() != b[i].tobytes()]
    print(tag, "frames that differ:", len(bad), bad[:12])
    if bad:
        i = bad[0]
        for name in a.dtype.names:
            if a[i][name].tobytes() != b[i][name].tobytes():
                print("   first differing frame", i, "field", name, np.asarray(a[i][name]).ravel()[:8], np.asarray(b[i][name]).ravel()[:8])
diff(out[50], out["again"], "batch 50 run 1 vs run 2:")
diff(out[50], out[8], "batch 50 vs batch 8:")
a = np.frombuffer(out[50].tobytes(), np.uint8).reshape(1000, -1); b = np.frombuffer(out["again"].tobytes(), np.uint8).reshape(1000, -1)
offs = np.nonzero((a != b).any(0))[0]
print("record size", a.shape[1], "differing byte offsets:", len(offs), offs[:40])
dt = out[50].dtype
for name in dt.names:
    o = dt.fields[name][1]; sz = dt.fields[name][0].itemsize
    hit = [int(x) for x in offs if o <= x < o + sz]
    if hit:
        print("  inside field", name, "offset", o, "size", sz, "->", hit[:10])
        sub = dt.fields[name][0]
        if sub.subdtype and sub.subdtype[0].names:
            el = sub.subdtype[0]; per = el.itemsize
            rel = sorted({(h - o) % per for h in hit})
            print("     element size", per, "relative offsets", rel[:16], "field map", {n: el.fields[n][1] for n in el.names})

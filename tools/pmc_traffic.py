"""Turn two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE; separate runs as MI355X_MICROARCH.md §HBM prescribes) of the
bench command into HBM bytes per convolution launch.  Corrections per the guide: counters are in KiB
(bytes = value * 1024) and on gfx950 FETCH_SIZE reports half of the bytes of a wide coalesced read stream (x2)."""
import csv, glob, hashlib, json, os, sys

def load(d, counter):
    f = sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True))[0]
    per = {}
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != counter:
            continue
        k = r["Kernel_Name"]
        per.setdefault(k, [0, 0.0])
        per[k][0] += 1
        per[k][1] += float(r["Counter_Value"])
    return per

fetch, write = load(sys.argv[1], "FETCH_SIZE"), load(sys.argv[2], "WRITE_SIZE")
_lib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "eagle_amd", "libeagle_hip.so")
out = {"build": "libeagle_hip.so md5 " + hashlib.md5(open(_lib, "rb").read()).hexdigest()[:12] if os.path.exists(_lib) else "?", "batch": 50, "detector": "n", "precision": (sys.argv[sys.argv.index("--precision") + 1] if "--precision" in sys.argv else "f32s"), "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) -- python3 bench.py " + " ".join(sys.argv[4:]),
       "corrections": "bytes = KiB*1024; FETCH_SIZE doubled (gfx950 under-report of wide coalesced reads)", "kernels": {}}
tf = tw = n = 0
for k in fetch:
    # the convolution family of the KEY-POINT network's precision (the default handle runs its detector in the exact fp32 family: those
    # conv_f32_kernel launches belong to `detector_convs`, not to the roofline family)
    fam = ("conv_f32_kernel",) if out["precision"] == "f32" else ("conv_f16", "conv_split", "bneck_split")      # (bneck_split_kernel: a whole Bottleneck of layer 1 as one launch, round 6)
    if not any(t in k for t in fam):
        continue
    c, v = fetch[k]
    w = write.get(k, [c, 0.0])[1]
    out["kernels"][k] = {"launches": c, "fetch_bytes_per_launch": 2 * v * 1024 / c, "write_bytes_per_launch": w * 1024 / c}
    tf += 2 * v * 1024; tw += w * 1024; n += c
out["conv_family"] = {"launches": n, "hbm_bytes_per_launch": (tf + tw) / max(n, 1), "fetch_bytes_per_launch": tf / max(n, 1),
                      "write_bytes_per_launch": tw / max(n, 1)}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out["conv_family"]))

"""Is the stateless path (conv networks) reproducible while something else runs on the GPU?  (developer probe)"""
import ctypes, os, sys, threading, time
import numpy as np
sys.path.insert(0, "."); sys.path.insert(0, "tests")
from eagle_amd import lib, synth, weights
from eagle_amd.coordinate_model import CoordinateModel
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhammer.so"))
hs, ys = weights.make_hrnet_state_dict(0), weights.make_yolo_state_dict("n", 0)
prec = os.environ.get("P_PREC", "f16")
if os.environ.get("P_FORCE"):
    os.environ["EAGLE_CONV_FORCE"] = os.environ["P_FORCE"]
A = CoordinateModel(precision=prec, batch=8, hrnet_state_dict=hs, detector_state_dict=ys)
os.environ.pop("EAGLE_CONV_FORCE", None)
frames = synth.clip(0, 8)
ref = A.process_records(frames).copy()
same = all(all(A.process_records(frames)[f].tobytes() == ref[f].tobytes() for f in ref.dtype.names) for _ in range(5))
print(f"[{prec} force={os.environ.get('P_FORCE')}] idle GPU: 5 repeated batches identical to the first: {same}", flush=True)
for other in sys.argv[1:] or ["none", "2", "0", "3"]:
    stop = False
    def busy():
        while not stop:
            if other != "none":
                assert L.hammer_launch(int(other), 2) == 0
            else:
                time.sleep(0.01)
    t = threading.Thread(target=busy); t.start(); time.sleep(0.2)
    n = bad = 0
    fields = {}
    for _ in range(30):
        r = A.process_records(frames)
        n += 1
        if any(r[f].tobytes() != ref[f].tobytes() for f in r.dtype.names):
            bad += 1
            for f in r.dtype.names:
                if r[f].tobytes() != ref[f].tobytes():
                    fields[f] = fields.get(f, 0) + 1
    stop = True; t.join()
    print(f"  other stream = hammer {other}: {bad} of {n} batches differ from the idle records; differing fields {fields}", flush=True)
A.handle.close()

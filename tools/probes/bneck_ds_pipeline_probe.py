"""Developer probe: default handle with block 0's downsample inside the fused launch (EAGLE_BNECK_DS=1) against =0 under several launch shapes."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eagle_amd import lib, synth, weights
from eagle_amd.coordinate_model import CoordinateModel
hs, ys = weights.make_hrnet_state_dict(0), weights.make_yolo_state_dict("n", 0)
frames = np.stack([synth.frame(0, t) for t in range(6)])
def run(ds, batch, nfr, **env):
    os.environ["EAGLE_BNECK_DS"] = ds
    for k, v in env.items(): os.environ[k] = v
    cm = CoordinateModel(batch=batch, hrnet_state_dict=hs, detector_state_dict=ys, allow_saturation=True, use_graph=0, multi_stream=0)
    r = cm.process_records(frames[:nfr])
    t = cm.handle.timings()
    cm.handle.close()
    for k in env: os.environ.pop(k)
    return r, t.sat_events
for batch, nfr in ((1, 1), (1, 3), (3, 3), (3, 6)):
    ref, _ = run("0", batch, nfr)
    for dsv, env in (("1", {}), ("1", {}), ("1", {"EAGLE_BNECK_WGS": "100000"}), ("1", {"EAGLE_BNECK_FORM": "1"})):
        r, sat = run(dsv, batch, nfr, **env)
        env = dict(env, DS=dsv)
        d = np.abs(r["hm_score"] - ref["hm_score"]).max(axis=1)
        print(f"batch {batch} frames {nfr} {env}: sat {sat}; per-frame max score diff {np.round(d, 4).tolist()}", flush=True)

import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
from eagle_amd import synth, weights
from oracle import pipeline
hs, ys = weights.make_hrnet_state_dict(0), weights.make_yolo_state_dict("n", 0)
m = pipeline.OracleModel(hs, ys, backend="c")
f = synth.frame(0, 0)
t = time.time(); m.step(f, 0); print("OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"), "step", round(time.time() - t, 1), "s")

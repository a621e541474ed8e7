"""Developer probe: frames/s of the resident pipeline with and without the per-frame geometry kernel (eagle_debug skip bit 8): how much the
RANSAC / projection kernel that overlaps the next batch on its own stream costs the convolutions it runs beside."""
import os, sys, time, json
os.environ["EAGLE_ENABLE_DEBUG"] = "1"
sys.path.insert(0, os.getcwd())
import numpy as np
from eagle_amd import lib, synth, weights
from eagle_amd.coordinate_model import CoordinateModel
B = 50
cm = CoordinateModel(batch=B)
h = cm.handle
base = synth.clip(seed=0, n=10, h=720, w=1280)
frames = np.concatenate([base] * (B * 10 // len(base)))
d = h.upload(frames)
def run(skip):
    lib.debug("skip", skip)
    h.process_device(d, len(frames))
    t = time.perf_counter()
    h.process_device(d, len(frames))
    return len(frames) / (time.perf_counter() - t)
for rep in range(2):
    for skip in (0, 8):
        print("skip", skip, round(run(skip), 1))
lib.debug("skip", 0)

"""Offline helper (run once in the build container): choose the synthetic detector's class-head biases so a
synthetic frame yields a realistic number of NMS candidates.  Prints the table that is pasted
into eagle_amd/weights.py::CLS_BIAS_TABLE.  Uses the oracle's torch backend; not part of the product."""
import sys

import numpy as np

from eagle_amd import synth, weights as W
from oracle import host, nets

variant = sys.argv[1] if len(sys.argv) > 1 else "n"
imgsz = int(sys.argv[2]) if len(sys.argv) > 2 else 640
seed = 0
# target number of anchors per class above the 0.15 floor, summed over the three levels
target = {0: 250, 1: 30, 2: 24, 3: 50, 4: 30}
frames = [synth.frame(seed, t) for t in (0, 37, 74)]
sd = W.make_yolo_state_dict(variant, seed, cls_bias=0.0, table=False)
for l in range(3):
    sd[f"model.22.cv3.{l}.2.bias"][:] = 0.0
zs = [[], [], []]
for f in frames:
    x, g = host.preprocess_detector(f, imgsz)
    for l, (b, c) in enumerate(nets.yolo_heads(sd, x, variant, backend="torch")):
        zs[l].append(c.reshape(-1, c.shape[-1]))
table = []
share = (0.6, 0.3, 0.1)
for l in range(3):
    z = np.concatenate(zs[l], 0)
    row = []
    for c in range(5):
        k = max(1, int(round(target[c] * share[l] * len(frames))))
        thr = np.sort(z[:, c])[-k]
        row.append(round(float(-1.7346 - thr), 4))
    table.append(row)
print((variant, seed), table)

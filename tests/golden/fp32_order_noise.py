"""How far apart are two legitimate fp32 evaluations of the same networks?  The oracle's exact-order C backend (fmaf chain in the canonical
K order) against its torch / oneDNN backend, same weights, same frame.  The spread is the yardstick for the split-precision family's
tolerances in tests/test_gpu_pipeline.py (_f32s_record_parity).  CPU only; ~1 minute.

Measured 2026-10-02 (8 vCPU container):
  cfg 2 (yolov8n@640 + HRNet-W48, 1280x720): heat-map indices 57/57 equal, scores 2.4e-7, logits 8.8e-7 of max|logit|, confidences 2.4e-6,
        267 detections on both sides with ONE pair of ids swapped (index-wise box deviation 784 px)
  cfg 3 (yolov8l@960, 1920x1080): decoded rows 4.4e-3 px, class confidences 2.0e-5, matched boxes 5.3e-3 px"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eagle_amd import synth, weights  # noqa: E402
from oracle import pipeline  # noqa: E402

hs, ys = weights.make_hrnet_state_dict(0), weights.make_yolo_state_dict("n", 0)
f = synth.frame(0, 0)
t = time.time()
a = pipeline.OracleModel(hs, ys, backend="c").step(f, 0)[1]
b = pipeline.OracleModel(hs, ys, backend="torch").step(f, 0)[1]
print(f"cfg2 ({time.time() - t:.0f} s): heat-map indices equal {(a['hm_idx'] == b['hm_idx']).sum()}/57, score dev {np.abs(a['hm_score'] - b['hm_score']).max():.3g}, "
      f"logit dev {np.abs(a['logits'] - b['logits']).max() / np.abs(a['logits']).max():.3g} of max")
n = min(len(a["dets"]), len(b["dets"]))
print(f"      detections {len(a['dets'])} / {len(b['dets'])}, index-wise conf dev {np.abs(a['dets'][:n, 4] - b['dets'][:n, 4]).max():.3g}, "
      f"index-wise box dev {np.abs(a['dets'][:n, :4] - b['dets'][:n, :4]).max():.3g} px")
yl = weights.make_yolo_state_dict("l", 0)
f3 = synth.frame(0, 3, 1080, 1920)
_, da, ra = pipeline.OracleModel(hs, yl, variant="l", imgsz=960, backend="c").detect_objects(f3)
_, db, rb = pipeline.OracleModel(hs, yl, variant="l", imgsz=960, backend="torch").detect_objects(f3)
dev = np.array([(abs(db[j, 4] - d[4]), np.abs(db[j, :4] - d[:4]).max()) for d in da for j in [np.abs(db[:, :4] - d[:4]).max(1).argmin()]])
print(f"cfg3: rows dev {np.abs(ra - rb).max():.3g}, class conf dev {np.abs(ra[:, 4:] - rb[:, 4:]).max():.3g}, matched conf dev {dev[:, 0].max():.3g}, matched box dev {dev[:, 1].max():.3g} px")

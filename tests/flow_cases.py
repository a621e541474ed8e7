"""Clips and canned network outputs for the optical-flow cadence fixtures (tests/golden/flow_golden.json).

Shared by tests/golden/make_golden.py::dump_flow (which drives the REFERENCE loop with them, in the build container) and
by the tests (which drive the oracle restatement and the HIP path with the same inputs).  Frames are regenerated from
seeds; pixel [0,0,0] carries the frame index so that stubbed networks can serve per-frame canned outputs."""
import numpy as np

from eagle_amd import synth

# name -> (fps, num_homography, num_keypoint_detection, [(seed, t) per frame], calibration)
CLIPS = {
    # keypoint_interval 3, homography_interval 6: plain propagation between detections
    "pan": (6, 1, 2, [(0, 3 * i) for i in range(13)], False),
    # frames 0 and 1 detect < 4 key-points: forward search + backward flow of cm.py:289-311
    "late_start": (6, 1, 2, [(1, 2 * i) for i in range(8)], False),
    # a scene cut on an unscheduled frame (flow collapses -> on-demand detection) and a sparse scheduled frame (flow merge)
    "cut": (8, 2, 2, [(0, 2 * i) for i in range(5)] + [(3, 40 + 2 * i) for i in range(7)], False),
    # reference default cadence at 25 fps (HRNet every 8th frame, H every 25th): long propagation chains, border losses
    "fps25": (25, 1, 3, [(2, 60 + 4 * i) for i in range(27)], False),
    # a black frame on an unscheduled index: every point loses its status -> empty flow -> on-demand detection (cm.py:316-319);
    # the sparse detection on the next unscheduled index after it keeps the chain short
    "blackout": (8, 2, 2, [(0, 0), (0, 2), (-1, 4), (0, 6), (0, 8), (0, 10), (-1, 12), (-1, 14), (0, 16), (0, 18)], False),
    # brightness calibration on
    "calib": (6, 1, 2, [(4, 3 * i) for i in range(7)], True),
    # a dark key-point on the image's first column: the reference indexes a 3-column grid at [3, 3] and raises IndexError (cm.py:545)
    "calib_edge": (6, 1, 2, [(4, 3 * i) for i in range(3)], True),
    # one-frame clip; a clip shorter than the key-point interval
    "single": (25, 1, 3, [(0, 0)], False),
    "short": (25, 1, 3, [(0, 2 * i) for i in range(3)], False),
    # no frame ever detects 4 key-points: the forward search runs off the end (cm.py:291-299), every later frame detects on demand
    "never4": (6, 1, 2, [(1, 2 * i) for i in range(5)], False),
}


def frames_of(name):
    fr = []
    for i, (seed, t) in enumerate(CLIPS[name][3]):
        f = synth.frame(seed, t).copy() if seed >= 0 else np.zeros((720, 1280, 3), np.uint8)
        f[0, 0, 0] = i
        fr.append(f)
    return fr


def canned(name):
    """-> (kps per frame: list of (heat-map index, x_n, y_n, score), dets per frame: (n,6) float32)"""
    spec = CLIPS[name][3]
    rng = np.random.default_rng(sum(map(ord, name)))
    kps, dets = [], []
    for i, (seed, t) in enumerate(spec):
        vis = synth.visible_landmarks(max(seed, 0), t)
        kp = []
        for idx, (x, y) in sorted(vis.items()):
            if rng.random() < 0.8:
                xn = min(239, int(round(x / 1280 * 239))) / 239
                yn = min(134, int(round(y / 720 * 134))) / 134
                kp.append((idx, xn, yn, float(np.float32(rng.uniform(0.31, 0.99)))))
        if name == "late_start" and i < 2:
            kp = kp[: 2 + i]
        if name == "cut" and i == 8:
            kp = kp[:2]
        if name == "blackout" and i in (6, 7):
            kp = kp[:3]
        if name == "never4":
            kp = kp[:3]
        if name == "calib_edge" and i == 0:
            kp = kp + [(56 if all(t[0] != 56 for t in kp) else 55, 0.0, 0.5, 0.9)]
        nd = int(rng.integers(0, 30))
        d = np.zeros((nd, 6), np.float32)
        d[:, 0] = rng.uniform(-5, 1250, nd); d[:, 1] = rng.uniform(-5, 690, nd)
        d[:, 2] = d[:, 0] + rng.uniform(4, 90, nd); d[:, 3] = d[:, 1] + rng.uniform(8, 160, nd)
        d[:, [0, 2]] = d[:, [0, 2]].clip(0, 1280); d[:, [1, 3]] = d[:, [1, 3]].clip(0, 720)
        d[:, 4] = np.sort(rng.uniform(0.15, 0.97, nd))[::-1]
        d[:, 5] = rng.choice([0, 0, 0, 0, 1, 2, 3, 4], nd)
        kps.append(kp); dets.append(d)
    return kps, dets

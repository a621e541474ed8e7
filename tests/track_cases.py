"""Synthetic detection sequences for the track-identity tests (tests/test_oracle_tracker.py, tests/test_gpu_tracker.py):
moving boxes with confidences and classes, as the detector stage would hand them to the tracker (descending confidence per frame)."""
import numpy as np

W, H = 1280, 720


def _box(cx, cy, w=36, h=88):
    return [cx - w / 2, cy - h / 2, cx + w / 2, cy + h / 2]


def make_clip(name, n=48):
    """-> list of float32 [k,6] arrays (x1,y1,x2,y2,conf,cls), rows sorted by descending confidence."""
    rng = np.random.default_rng(sum(map(ord, name)))
    frames = []
    for f in range(n):
        rows = []

        def add(cx, cy, conf, cls=0, w=36, h=88, jitter=1.5):
            b = _box(cx + rng.normal(0, jitter), cy + rng.normal(0, jitter), w + rng.normal(0, 1), h + rng.normal(0, 1))
            rows.append(b + [conf, cls])
        if name == "parallel":                      # eight players walking in parallel lanes, a goalkeeper, a referee and the ball
            for k in range(8):
                add(80 + 14 * f + 6 * k, 90 + 70 * k, 0.62 + 0.04 * k)
            add(1180 - 2 * f, 360, 0.9, cls=1)
            add(640 + 3 * f, 200, 0.8, cls=3)
            add(300 + 4 * f, 500, 0.7, cls=2, w=14, h=14, jitter=0.5)
        elif name == "crossing":                    # two pairs cross each other at different depths
            add(200 + 16 * f, 300, 0.9); add(968 - 16 * f, 318, 0.85)
            add(300 + 9 * f, 520, 0.8); add(900 - 11 * f, 560, 0.75)
        elif name == "occlusion":                   # detections drop out for 6 and for 40 frames
            add(150 + 8 * f, 250, 0.9)
            if not 10 <= f < 16:
                add(400 + 5 * f, 400, 0.88)
            if f < 4 or f >= 44:
                add(900 - 2 * f, 550, 0.86)
            add(640, 120 + 6 * f, 0.7, cls=1)
        elif name == "births":                      # objects enter one after the other, some leave
            for k in range(6):
                if 5 * k <= f < 5 * k + 30:
                    add(100 + 180 * k + 4 * (f - 5 * k), 200 + 60 * (k % 3), 0.65 + 0.05 * (k % 4))
        elif name == "lowconf":                     # confidences dip below the high threshold (second association) and below new_track_thresh
            add(200 + 10 * f, 300, 0.9 if f % 7 else 0.32)
            add(800 - 7 * f, 420, 0.55)             # never above new_track_thresh 0.6: never starts a track -> raw-detection fallback frames
            add(500, 100 + 9 * f, 0.8 if f < 20 else 0.2)
        elif name == "crowd":                       # a dense cluster with overlapping boxes, noisier detections
            for k in range(12):
                add(400 + 45 * (k % 4) + 5 * f * ((k % 3) - 1), 250 + 60 * (k // 4) + 3 * f * ((k % 2) * 2 - 1), 0.6 + 0.03 * k, jitter=3.0)
        else:
            raise KeyError(name)
        a = np.array(rows, np.float32).reshape(-1, 6)
        frames.append(a[np.argsort(-a[:, 4], kind="stable")])
    return frames


CLIPS = ["parallel", "crossing", "occlusion", "births", "lowconf", "crowd"]


def pan(clip, dx=42.0, dy=-9.0):
    """The same detections seen by a camera that pans by (dx, dy) pixels per frame, and the per-frame warps a camera-motion estimator would
    report (frame i-1 -> i; identity for frame 0): without compensation the constant-velocity filter has to absorb a jump of a box width."""
    out, warps = [], []
    for i, d in enumerate(clip):
        e = d.copy()
        e[:, [0, 2]] += np.float32(dx * i); e[:, [1, 3]] += np.float32(dy * i)
        out.append(e)
        warps.append([1.0, 0.0, dx if i else 0.0, 0.0, 1.0, dy if i else 0.0])
    return out, np.array(warps, np.float64)

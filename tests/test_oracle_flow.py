"""Optical-flow cadence: the oracle's restatement of the reference loop against the records the REFERENCE's
own loop produced on the same clips and canned network outputs (tests/golden/flow_golden.json, made by
tests/golden/make_golden.py::dump_flow), plus known answers for the cv2 restatements of oracle/eo_flow.c."""
import json
import os

import numpy as np
import pytest

import flow_cases
from oracle import flow, host
from oracle import prims as P

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "flow_golden.json")))


def _norm(o):
    return json.loads(json.dumps(o, default=lambda v: v.tolist() if isinstance(v, np.ndarray) else (int(v) if isinstance(v, np.integer) else float(v))))


def oracle_loop(name):
    fps, nh, nk, spec, calib = flow_cases.CLIPS[name]
    frames = flow_cases.frames_of(name)
    kps, dets = flow_cases.canned(name)
    h, w = frames[0].shape[:2]
    return flow.loop_records(frames, fps, nh, nk,
                             lambda i: host.keypoints_from_decoded([t for t in kps[i] if t[3] > 0.01], h, w, 0.3),
                             lambda i: host.objects_from_detections(dets[i], h, w, 0.35), calibration=calib)


@pytest.mark.parametrize("name", sorted(flow_cases.CLIPS))
def test_loop_restatement_equals_reference_loop(name):
    gold = GOLD[name]
    if gold["raises"]:
        with pytest.raises(Exception) as e:
            oracle_loop(name)
        assert type(e.value).__name__ == gold["raises"]
        return
    res, stats = oracle_loop(name)
    assert sorted(set(stats["detect_calls"])) == gold["detected_frames"]
    for i in range(len(res)):
        got, ref = _norm(res[i]), gold["records"][str(i)]
        for cls in got["Coordinates"].values():
            for o in cls.values():
                o.pop("_pitch_float", None)
        assert got["Keypoints"] == ref["Keypoints"], (name, i)
        assert list(got["Keypoints"]) == list(ref["Keypoints"]), (name, i, "dict order")
        assert got["Coordinates"] == ref["Coordinates"], (name, i)
        assert got["Boundaries"] == ref["Boundaries"] and got["Time"] == ref["Time"], (name, i)


def test_gray_hsv_known_answers():
    px = np.array([[[255, 0, 0], [0, 255, 0], [0, 0, 255], [255, 255, 255], [0, 0, 0], [10, 200, 100], [128, 128, 128], [1, 2, 3]]], np.uint8)
    # BGR2GRAY: Y = (3735 B + 19235 G + 9798 R + 2^14) >> 15
    assert P.bgr2gray(px).tolist() == [[29, 150, 76, 255, 0, 148, 128, 2]]
    hsv = P.bgr2hsv(px)[0]
    assert hsv[:5].tolist() == [[120, 255, 255], [60, 255, 255], [0, 255, 255], [0, 0, 255], [0, 0, 0]]
    # against the float definition, +-1 from the table rounding
    rng = np.random.default_rng(0)
    c = rng.integers(0, 256, (1, 4000, 3), dtype=np.uint8)
    got = P.bgr2hsv(c)[0].astype(int)
    b, g, r = (c[0, :, k].astype(float) for k in range(3))
    v = np.maximum(np.maximum(b, g), r); mn = np.minimum(np.minimum(b, g), r); d = v - mn
    hh = np.where(d == 0, 0, np.where(v == r, 60 * (g - b) / np.where(d == 0, 1, d), np.where(v == g, 120 + 60 * (b - r) / np.where(d == 0, 1, d), 240 + 60 * (r - g) / np.where(d == 0, 1, d))))
    hh = np.where(hh < 0, hh + 360, hh) / 2
    dh = np.abs(got[:, 0] - hh); dh = np.minimum(dh, 180 - dh)
    assert dh.max() <= 1.0 and np.array_equal(got[:, 2], v.astype(int))
    assert np.abs(got[:, 1] - np.where(v == 0, 0, 255 * d / np.where(v == 0, 1, v))).max() <= 1.0


def test_pyrdown_matches_float_gaussian():
    rng = np.random.default_rng(1)
    g = rng.integers(0, 256, (37, 52), dtype=np.uint8)
    d = P.pyrdown(g)
    assert d.shape == (19, 26)
    k = np.array([1, 4, 6, 4, 1], float) / 16
    pad = np.pad(g.astype(float), 2, mode="reflect")
    full = sum(k[a] * k[b] * pad[a:a + 37, b:b + 52] for a in range(5) for b in range(5))
    assert np.abs(d - full[::2, ::2]).max() <= 0.5 + 1e-9


def test_lk_recovers_a_known_translation():
    rng = np.random.default_rng(2)
    base = rng.integers(0, 256, (40, 60)).astype(float)
    big = np.kron(base, np.ones((8, 8)))                               # 320 x 480, blocky texture
    from scipy.ndimage import gaussian_filter, shift
    big = gaussian_filter(big, 2.0)
    a = np.clip(big, 0, 255).astype(np.uint8)
    b = np.clip(shift(big, (2.25, -3.5), order=3, mode="reflect"), 0, 255).astype(np.uint8)
    pts = np.array([[100, 100], [240, 160], [300.5, 90.25], [60, 250]], np.float32)
    nxt, st = P.calc_optical_flow_pyr_lk(a, b, pts)
    assert st.all()
    assert np.abs((nxt - pts) - np.array([-3.5, 2.25])).max() < 0.1
    # a point whose window leaves the frame loses its status
    nxt, st = P.calc_optical_flow_pyr_lk(a, b, np.array([[-30, 10], [5000, 10]], np.float32))
    assert not st.any()

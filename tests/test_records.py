"""CPU test of the host-side record -> reference-dict conversion (eagle_amd/records.py) against the records the
reference's own loop body produced (tests/golden/loop_golden.json): the EagleFrameResult is filled here from the
oracle's intermediate values exactly the way the device kernels fill it."""
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
CASES = json.load(open(os.path.join(HERE, "golden", "loop_golden.json")))


def _canon(d):
    if isinstance(d, dict):
        return {str(k): _canon(v) for k, v in d.items() if not str(k).startswith("_")}
    if isinstance(d, (list, tuple)):
        return [_canon(v) for v in d]
    if isinstance(d, np.integer):
        return int(d)
    if isinstance(d, np.floating):
        return float(d)
    return d


def _fill_record(c):
    from eagle_amd.lib import RESULT_DTYPE
    from eagle_amd.pitch import PITCH_POINTS_TO_INTERSECTION, on_plane_mask
    from oracle import host, prims as P
    rec = np.zeros(1, RESULT_DTYPE)[0]
    dets = np.array(c["dets"], np.float32).reshape(-1, 6)
    decoded = [(int(i), x, y, s) for i, x, y, s in c["kp"] if s > 0.01]
    kps = host.keypoints_from_decoded(decoded, 720, 1280)
    det_labels = set(kps)
    if len(kps) >= 2:
        kps = host.synthesize_keypoints(kps)
    H, kept = host.solve_homography(kps)
    plane = on_plane_mask()
    rec["n_kp"] = len(kps)
    for k, (lab, (x, y)) in enumerate(kps.items()):
        i = PITCH_POINTS_TO_INTERSECTION[lab]
        rec["kp"][k]["label"], rec["kp"][k]["x"], rec["kp"][k]["y"] = i, x, y
        rec["kp"][k]["synthesized"] = lab not in det_labels
        rec["kp"][k]["on_plane"] = plane[i]
        rec["kp"][k]["inlier"] = H is not None and lab in kept
    rec["H_valid"] = H is not None
    if H is not None:
        rec["H"] = H.reshape(9)
    b = host.boundaries(H, 720, 1280)
    rec["bounds_valid"] = b[0] is not None
    if b[0] is not None:
        rec["bounds"] = [b[0][0], b[1][0], b[2][0], b[3][0]]
    rec["n_det"] = len(dets)
    ball = 0
    for k, d in enumerate(dets):
        r = rec["det"][k]
        r["x1"], r["y1"], r["x2"], r["y2"], r["conf"], r["cls"] = d[0], d[1], d[2], d[3], d[4], int(d[5])
        ib = d[:4].astype(int)
        cls = int(d[5])
        r["id"] = -1
        if cls in (0, 1):
            ib = [min(max(ib[0], 0), 1279), min(max(ib[1], 0), 719), min(max(ib[2], 0), 1279), min(max(ib[3], 0), 719)]
            r["id"], r["reported"] = k, not (float(d[4]) < 0.35)
        elif cls == 2:
            r["id"], r["reported"] = ball, not (float(d[4]) < 0.35)
            ball += 1
        r["bx1"], r["by1"], r["bx2"], r["by2"] = ib
        r["foot_x"], r["foot_y"] = (int(ib[0]) + int(ib[2])) // 2, ib[3]
        if H is not None:
            tf = P.perspective_transform(np.array([[r["foot_x"], r["foot_y"]]], np.float32), H)
            tx, ty = int(tf[0, 0]), int(tf[0, 1])
            r["pitch_xf"], r["pitch_yf"], r["pitch_x"], r["pitch_y"] = tf[0, 0], tf[0, 1], tx, ty
            r["in_bounds"] = not (tx < 0 or tx > 105 or ty < 0 or ty > 68)
    return rec


@pytest.mark.parametrize("ci", range(len(CASES)))
def test_record_to_reference_dict(ci):
    from eagle_amd import records
    rec = _fill_record(CASES[ci])
    assert _canon(records.to_reference_dict(rec, 0, 1)) == CASES[ci]["record"]


def test_process_dict_view():
    from eagle_amd import records
    out = records.to_process_dict(_fill_record(CASES[0]))
    assert set(out) >= {"players", "ball", "H"} and out["H"].shape == (3, 3)
    assert all(v["Type"] in ("Player", "Goalkeeper") for v in out["players"].values())


def test_ingest_sampling_equals_reference_read_video():
    """eagle_amd/io.py against the indices the reference's own read_video kept (tests/golden/ingest_golden.json: the reference
    function run over a stubbed cv2.VideoCapture in the build container)."""
    import json
    import os
    import pytest
    from eagle_amd import io as eio
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ingest_golden.json")))
    for g in gold:
        if g["raises"]:
            with pytest.raises(Exception) as e:
                eio.sample_indices(g["n"], g["native_fps"], g["fps"])
            assert type(e.value).__name__ == g["raises"]
        else:
            assert eio.sample_indices(g["n"], g["native_fps"], g["fps"]) == g["kept"]
    clip = np.arange(12 * 2 * 2 * 3, dtype=np.uint8).reshape(12, 2, 2, 3)
    frames, fps = eio.read_clip(clip, 50.0, 24)
    assert fps == 24 and frames.shape == (6, 2, 2, 3) and np.array_equal(frames, clip[::2])


def test_coordinate_model_passes_the_graph_mode_through():
    """EagleConfig::use_graph takes 0, 1 and 2 (replay only inside calls of >= 3 steps); the wrapper used to map everything through bool()."""
    import pytest
    from eagle_amd import lib
    from eagle_amd.coordinate_model import CoordinateModel
    gm = CoordinateModel._graph_mode
    assert gm(None) == lib.AUTO and gm(False) == 0 and gm(True) == 1 and gm(0) == 0 and gm(1) == 1 and gm(2) == 2
    with pytest.raises(ValueError):
        gm(3)

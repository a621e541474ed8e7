"""-m gpu: track identities through the C ABI (eagle_track_open / eagle_track_frames) against oracle/tracker.py on six synthetic
detection sequences (parallel walkers, crossings, occlusion gaps, births / exits, low-confidence dips, a crowd): per frame the
reference-shaped object dict (cm.py:577-627: ids, integer boxes, foot points, the detection-index fallback on frames without a
tracked player) and the pitch coordinates of the smoothed foot points (re-projected on the GPU with the frame's homography)."""
import numpy as np
import pytest

import track_cases
from eagle_amd import lib, records
from oracle import host
from oracle.tracker import BotSortLite, objects_from_tracks

pytestmark = pytest.mark.gpu

H_TEST = np.array([[0.075, 0.012, -8.0], [0.004, 0.118, -9.0], [2e-5, 3.1e-4, 1.0]])


def _records(clip):
    """What the detector stage leaves in the records (cm.py:598-627 rules for the raw detections), with one homography per frame."""
    recs = np.zeros(len(clip), lib.RESULT_DTYPE)
    for i, d in enumerate(clip):
        r = recs[i]
        n = len(d)
        r["n_det"] = n
        r["H_valid"] = 1; r["H"] = H_TEST.ravel()
        obj = host.objects_from_detections(d, track_cases.H, track_cases.W)
        ball_k = 0
        for k in range(n):
            e = r["det"][k]
            e["x1"], e["y1"], e["x2"], e["y2"], e["conf"], e["cls"] = d[k]
            cls = int(d[k, 5])
            e["id"] = -1
            if cls in (0, 1):
                name = "Player" if cls == 0 else "Goalkeeper"
                if k in obj[name]:
                    o = obj[name][k]
                    e["id"] = k; e["reported"] = 1
                    e["bx1"], e["by1"], e["bx2"], e["by2"] = o["BBox"]; e["foot_x"], e["foot_y"] = o["Bottom_center"]
            elif cls == 2:
                if ball_k in obj.get("Ball", {}):
                    o = obj["Ball"][ball_k]
                    e["id"] = ball_k; e["reported"] = 1
                    e["bx1"], e["by1"], e["bx2"], e["by2"] = o["BBox"]; e["foot_x"], e["foot_y"] = o["Bottom_center"]
                ball_k += 1
    return recs


def test_clip_motion_equals_oracle():
    """eagle_clip_motion (gray pyramids + the key-point LK kernel on the 8 x 6 grid + similarity RANSAC on the host side of the library) against
    oracle/tracker.py::camera_motion on frames of the panning synthetic camera, incl. a pair without motion and a scene cut."""
    from eagle_amd import synth
    from oracle.tracker import camera_motion
    from eagle_amd.coordinate_model import CoordinateModel
    frames = [synth.frame(0, t) for t in (4, 6, 8, 8, 9)] + [synth.frame(1, 40)]
    h = CoordinateModel(batch=1).handle                    # (the clip session needs a finalised handle)
    d = h.upload(np.stack(frames))
    try:
        h.clip_open(d, len(frames))
        w = h.clip_motion(0, len(frames))
        w2 = h.clip_motion(2, 3)
        h.clip_close()
    finally:
        h.free(d); h.close()
    assert np.array_equal(w[0], [1, 0, 0, 0, 1, 0]) and np.array_equal(w[2:5], w2)
    for i in range(1, len(frames)):
        assert np.allclose(w[i].reshape(2, 3), camera_motion(frames[i - 1], frames[i]), rtol=0, atol=1e-9), i
    assert np.abs(w[3] - [1, 0, 0, 0, 1, 0]).max() < 1e-6          # identical frames: identity


@pytest.mark.parametrize("name,panned", [(n, False) for n in track_cases.CLIPS] + [("parallel", True), ("crossing", True)])
def test_track_ids_equal_oracle(name, panned):
    clip = track_cases.make_clip(name)
    warps = None
    if panned:                                              # a camera pan of a box width per frame, compensated with the warps (eagle_track_frames_cmc)
        clip, warps = track_cases.pan(clip)
        clip = [np.clip(d, -4000, 8000) for d in clip]
    h = lib.Handle(batch=1)
    recs = _records(clip)
    h.reproject(recs, np.tile(H_TEST.ravel(), (len(recs), 1)), np.ones(len(recs), np.uint8))      # the projection the geometry kernel would have done
    h.track_open()
    # two chunks: the tracker state carries across calls
    h.track_frames(recs[:20], None if warps is None else warps[:20]); h.track_frames(recs[20:], None if warps is None else warps[20:])
    h.close()
    tr = BotSortLite()
    tracked_frames = 0
    for i, d in enumerate(clip):
        out = tr.update(d, None if warps is None else warps[i].reshape(2, 3))
        obj = objects_from_tracks(out, track_cases.H, track_cases.W)
        if len(obj["Player"]) == 0 and len(obj["Goalkeeper"]) == 0:          # cm.py:598: fall back to the raw detections
            obj = host.objects_from_detections(d, track_cases.H, track_cases.W)
        else:
            raw = host.objects_from_detections(d, track_cases.H, track_cases.W)
            if "Ball" in raw:
                obj["Ball"] = raw["Ball"]
            tracked_frames += 1
        exp = host.project_objects(obj, H_TEST)
        got = records.to_reference_dict(recs[i], i)["Coordinates"]
        for cname in ("Player", "Goalkeeper", "Ball"):
            g, e = got.get(cname, {}), exp.get(cname, {})
            assert set(g) == set(e), (name, i, cname, sorted(g), sorted(e))
            for oid in e:
                assert [int(v) & 0xFFFF for v in e[oid]["BBox"]] == g[oid]["BBox"], (name, i, cname, oid)
                assert abs(e[oid]["Confidence"] - g[oid]["Confidence"]) < 1e-7
                assert e[oid]["Transformed_Coordinates"] == g[oid]["Transformed_Coordinates"], (name, i, cname, oid)
                assert e[oid].get("Image_Bottom_center") == g[oid].get("Image_Bottom_center")
    assert tracked_frames >= len(clip) - 6          # ("lowconf" starts with frames on which no track is confirmed yet: the raw-detection fallback)


def test_coordinate_model_with_tracker_and_camera_motion_runs_both_cadences():
    """CoordinateModel(tracker=True, camera_motion=True): the stateless route (own clip session for the motion) and the flow cadence (motion taken
    from the loop's session) produce the reference-shaped dict; with camera_motion the ids are still integers keyed per frame."""
    from eagle_amd import synth
    from eagle_amd.coordinate_model import CoordinateModel
    frames = np.stack([synth.frame(0, t) for t in range(6)])
    cm = CoordinateModel(batch=2, tracker=True, camera_motion=True, detector_conf=0.2)
    a = cm.get_coordinates(frames, fps=1)                                   # key-points on every frame: stateless route
    ids_a = {k for i in a for k in a[i]["Coordinates"].get("Player", {})}
    b = cm.get_coordinates(frames, fps=24, num_keypoint_detection=3)        # main.py's cadence: clip session
    ids_b = {k for i in b for k in b[i]["Coordinates"].get("Player", {})}
    # one tracker for the model's lifetime, like the reference's single BotSort (cm.py:66-72): the second clip continues the first one's
    # tracks (same frames -> same objects -> ids from the first clip reappear), and reset_tracker() starts from id 1 again
    cm.reset_tracker()
    c = cm.get_coordinates(frames, fps=1)
    cm.handle.close()
    assert c == a
    if ids_a and ids_b:
        assert ids_a & ids_b or min(ids_b) > max(ids_a)
    for res in (a, b):
        assert sorted(res) == list(range(6))
        for i in res:
            assert set(res[i]) >= {"Coordinates", "Time", "Keypoints", "Boundaries"}
            assert all(isinstance(k, int) for k in res[i]["Coordinates"].get("Player", {}))

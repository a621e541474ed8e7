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


@pytest.mark.parametrize("name", track_cases.CLIPS)
def test_track_ids_equal_oracle(name):
    clip = track_cases.make_clip(name)
    h = lib.Handle(batch=1)
    recs = _records(clip)
    h.reproject(recs, np.tile(H_TEST.ravel(), (len(recs), 1)), np.ones(len(recs), np.uint8))      # the projection the geometry kernel would have done
    h.track_open()
    # two chunks: the tracker state carries across calls
    h.track_frames(recs[:20]); h.track_frames(recs[20:])
    h.close()
    tr = BotSortLite()
    tracked_frames = 0
    for i, d in enumerate(clip):
        out = tr.update(d)
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

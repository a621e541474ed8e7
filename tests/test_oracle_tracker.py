"""CPU: known-answer tests of the track-identity restatement (oracle/tracker.py; BoT-SORT association with ReID / CMC off)."""
import numpy as np

import track_cases
from oracle.tracker import BotSortLite, objects_from_tracks


def _ids(name):
    tr = BotSortLite()
    return [{int(r[7]): int(r[4]) for r in tr.update(d)} for d in track_cases.make_clip(name)]      # per frame: detection index -> track id


def test_parallel_walkers_keep_their_ids():
    clip = track_cases.make_clip("parallel")
    tr = BotSortLite()
    first = None
    for f, d in enumerate(clip):
        out = tr.update(d)
        ids = sorted(int(r[4]) for r in out)
        if f == 0:
            first = ids
            assert len(ids) == 11          # every confident detection starts an activated track on the very first frame (ball and referee too)
        assert ids == first, f


def test_new_tracks_are_confirmed_one_frame_later_and_ids_increase():
    per = _ids("births")
    seen = []
    for f, m in enumerate(per):
        for i in m.values():
            if i not in seen:
                seen.append(i)
    assert seen == sorted(seen) and len(seen) == 6
    assert len(per[5]) == 1 and len(per[6]) == 2      # the object that enters on frame 5 is unconfirmed there, output from frame 6 on


def test_short_occlusion_keeps_the_id_long_occlusion_gets_a_new_one():
    clip = track_cases.make_clip("occlusion")
    tr = BotSortLite()
    by_frame = []
    for d in clip:
        out = tr.update(d)
        by_frame.append({int(r[4]) for r in out})
    at = lambda f: by_frame[f]
    assert at(9) == {1, 2, 4} and at(12) == {1, 4} and at(16) == {1, 2, 4}          # the 6-frame gap (frames 10-15) did not cost track 2 its id
    assert at(3) == {1, 2, 3, 4} and at(44) == {1, 2, 4} and at(45) == {1, 2, 4, 5}  # gone for 40 frames > track_buffer 30: a new id, one frame after it reappears


def test_low_confidence_detections_continue_tracks_but_never_start_one():
    clip = track_cases.make_clip("lowconf")
    tr = BotSortLite()
    for f, d in enumerate(clip):
        out = tr.update(d)
        for r in out:
            assert d[int(r[7]), 4] != np.float32(0.55), "a 0.55-confidence detection must not create a track (new_track_thresh 0.6)"
        if f > 0 and f % 7 == 0 and f < 40:
            dipped = [r for r in out if abs(d[int(r[7]), 4] - 0.32) < 1e-6]
            assert len(dipped) == 1 and int(dipped[0][4]) == 2, "the dipped detection continues its track (id 2) through the second association"


def test_objects_follow_cm_577_596():
    clip = track_cases.make_clip("parallel")
    tr = BotSortLite()
    out = tr.update(clip[0])
    obj = objects_from_tracks(out, track_cases.H, track_cases.W)
    assert set(obj) == {"Player", "Goalkeeper"} and len(obj["Player"]) == 8 and len(obj["Goalkeeper"]) == 1
    for tid, o in obj["Player"].items():
        x1, y1, x2, y2 = o["BBox"]
        assert o["Bottom_center"] == [int((x1 + x2) / 2), y2] and 0 <= x1 <= x2 < track_cases.W


def test_similarity_ransac_recovers_a_known_camera_motion():
    """oracle/tracker.py::similarity_ransac: rotation + zoom + shift under pixel noise and 20 % gross outliers."""
    import numpy as np
    from oracle import tracker as T
    rng = np.random.default_rng(0)
    p0 = rng.uniform(0, 1000, (40, 2)); th, sc = 0.01, 1.02
    R = np.array([[sc * np.cos(th), -sc * np.sin(th)], [sc * np.sin(th), sc * np.cos(th)]]); t = np.array([12.5, -7.25])
    p1 = p0 @ R.T + t + rng.normal(0, 0.2, p0.shape)
    p1[:8] += rng.uniform(-80, 80, (8, 2))
    W = T.similarity_ransac(p0, p1)
    assert np.abs(W[:, :2] - R).max() < 2e-4 and np.abs(W[:, 2] - t).max() < 0.15
    assert np.array_equal(T.similarity_ransac(p0[:1], p1[:1]), np.array([[1.0, 0, 0], [0, 1.0, 0]]))


def test_camera_motion_follows_the_synthetic_camera():
    """camera_motion (sparse LK on the 8 x 6 grid + RANSAC) between two frames of the panning synthetic camera: the warp maps the projections of pitch
    points at time t onto their projections at time t + 2 to within a pixel or two."""
    import numpy as np
    from eagle_amd import synth
    from oracle import tracker as T
    a, b = synth.frame(0, 4), synth.frame(0, 6)
    W = T.camera_motion(a, b)
    world = np.array([[x, y] for x in (20.0, 40.0, 52.5, 65.0, 85.0) for y in (10.0, 34.0, 58.0)])
    pa, pb = synth.project(synth.camera(0, 4), world), synth.project(synth.camera(0, 6), world)
    inside = (pa[:, 0] > 0) & (pa[:, 0] < 1280) & (pa[:, 1] > 0) & (pa[:, 1] < 720)
    pred = pa[inside] @ W[:, :2].T + W[:, 2]
    assert inside.sum() >= 6 and np.abs(pred - pb[inside]).max() < 3.0, np.abs(pred - pb[inside]).max()


def test_camera_motion_compensation_keeps_identities_under_a_fast_pan():
    """A pan of a box width per frame: with the warps every object keeps ONE id for the whole clip; without them tracks are lost and re-born."""
    import track_cases
    from oracle.tracker import BotSortLite
    clip, warps = track_cases.pan(track_cases.make_clip("parallel", 30))

    def ids_per_object(use_warp):
        tr = BotSortLite(); seen = {}
        for i, d in enumerate(clip):
            for row in tr.update(d, warps[i].reshape(2, 3) if use_warp else None):
                seen.setdefault(int(row[7]), set()).add(int(row[4]))        # detection index (stable per object in this clip) -> ids
        return seen
    with_cmc, without = ids_per_object(True), ids_per_object(False)
    assert all(len(v) == 1 for v in with_cmc.values()) and len(with_cmc) >= 9
    assert sum(len(v) for v in without.values()) > sum(len(v) for v in with_cmc.values())


def test_appearance_fusion_decides_where_iou_cannot():
    """BoT-SORT with_reid (oracle/tracker.py::_fuse_appearance): cost = min(IoU cost, embedding distance / 2), the embedding term only where
    it is below appearance_thresh 0.25 and the boxes are within proximity_thresh 0.5.  Two tracks and two detections with the SAME IoU cost
    for every pairing: the fused matrix prefers the pairs whose embeddings agree, and the assignment follows."""
    from oracle.tracker import _assign, _fuse_appearance, _Track
    fa, fb = np.zeros(512), np.zeros(512)
    fa[:256] = 1.0; fb[256:] = 1.0
    fa /= np.linalg.norm(fa); fb /= np.linalg.norm(fb)
    ta, tb, da, db = (_Track([0, 0, 10, 10, 0.9, 0], k) for k in range(4))
    ta.smooth_feat, tb.smooth_feat = fa, fb
    db.curr_feat, da.curr_feat = fa, 0.98 * fb + 0.02 * fa          # detection 0 looks like track b, detection 1 like track a
    iou = np.full((2, 2), 0.4)
    fused = _fuse_appearance(iou, iou, [ta, tb], [da, db])
    assert fused[0, 1] == 0.0 and fused[1, 0] < 1e-3 and fused[0, 0] == 0.4 and fused[1, 1] == 0.4
    m, _, _ = _assign(fused, 0.8)
    assert sorted(m) == [(0, 1), (1, 0)]
    # thresholds: a far embedding (distance / 2 > 0.25) or a far box (IoU distance > 0.5) leaves the IoU cost untouched
    far = np.full((2, 2), 0.6)
    assert np.array_equal(_fuse_appearance(far, far, [ta, tb], [da, db]), far)
    # pairs without both features keep the IoU cost
    da.curr_feat = None
    assert _fuse_appearance(iou, iou, [ta, tb], [da, db])[:, 0].tolist() == [0.4, 0.4]

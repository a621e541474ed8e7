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

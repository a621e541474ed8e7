"""-m gpu: the appearance branch of the tracker (the reference's BotSort runs with OSNet-x0.25 ReID, cm.py:66-72, 577).
(1) eagle_reid_features (crop + resize + normalise + OSNet-x0.25 on the GPU) against oracle/reid.py (numpy restatement, parity unpinned);
(2) eagle_track_frames_reid against oracle/tracker.py with the same embeddings: ids through a crossing of two players whose boxes coincide —
    appearance is the only thing that tells them apart — and unchanged behaviour without embeddings."""
import numpy as np
import pytest

from eagle_amd import lib, osnet
from oracle import reid
from oracle.tracker import BotSortLite

pytestmark = pytest.mark.gpu
H, W = 720, 1280
COLORS = [(40, 40, 220), (220, 60, 40), (30, 200, 230), (240, 240, 240)]          # BGR jerseys


def _handle():
    from eagle_amd import weights
    h = lib.Handle(batch=1)
    weights.load_into(h, [weights.make_hrnet_state_dict(0), weights.make_yolo_state_dict("n", 0), osnet.make_osnet_state_dict(0)])
    return h


def _render(boxes_per_frame, colors):
    """frames with a textured green pitch and one jersey-coloured, striped rectangle per box (later boxes drawn on top)"""
    rng = np.random.default_rng(3)
    frames = []
    for boxes in boxes_per_frame:
        f = np.empty((H, W, 3), np.uint8)
        f[:] = (60, 140, 50)
        f = np.clip(f.astype(np.int16) + rng.integers(-12, 13, f.shape), 0, 255).astype(np.uint8)
        for k, (x1, y1, x2, y2) in enumerate(boxes):
            x1, y1, x2, y2 = int(x1), int(y1), int(x2), int(y2)
            f[y1:y2, x1:x2] = colors[k]
            f[y1 + (y2 - y1) // 3: y1 + (y2 - y1) // 3 + 4 + 3 * k, x1:x2] = (20, 20, 20)        # a stripe whose width differs per player
        frames.append(f)
    return np.stack(frames)


def test_reid_features_equal_the_torch_oracle():
    from eagle_amd import synth
    sd = osnet.make_osnet_state_dict(0)
    views = ((0, 3), (1, 8), (2, 5), (0, 20))
    frames = np.stack([synth.frame(*v) for v in views])
    crops = []
    for i, (seed, t) in enumerate(views):
        for _, x0, y0, x1, y1 in synth.player_boxes(seed, t):
            r = reid.crop_box((x0, y0, x1, y1), H, W)
            if r is not None:
                crops.append((i, *r))
    crops += [(0, 0, 0, 128, 256), (1, 100, 50, 356, 562), (0, 1200, 600, 1279, 719), (1, 5, 5, 7, 9)]      # identity / exact 2x / border / tiny crops
    h = _handle()
    d = h.upload(frames)
    got = h.reid_features(d, len(frames), np.array(crops, np.int32))
    h.free(d); h.close()
    ref = reid.embed(sd, np.stack([reid.prepare_crop(frames[c[0]], c[1:]) for c in crops]))
    assert got.shape == ref.shape == (len(crops), 512) and len(crops) > 64          # more than one pass of 64 crops
    scale = np.abs(ref).max()
    assert np.abs(got - ref).max() <= 2e-4 * scale, np.abs(got - ref).max() / scale
    gn, rn = got / np.linalg.norm(got, axis=1, keepdims=True), ref / np.linalg.norm(ref, axis=1, keepdims=True)
    assert np.abs((gn * rn).sum(1) - 1).max() < 1e-6                                 # cosine similarity GPU vs oracle per crop


def _crossing_clip(n=45):
    """two players walk towards each other along one line, stand in the SAME box for five frames (identical detections: nothing but
    appearance — and for those frames not even that — tells them apart) and then turn back where they came from, which a constant-velocity
    model does not expect; a third stands apart"""
    dets, boxes = [], []
    for t in range(n):
        if t < 15:
            xa, xb = 300 + 10 * t, 600 - 10 * t
        elif t < 20:
            xa = xb = 450
        else:
            xa, xb = 450 - 10 * (t - 19), 450 + 10 * (t - 19)
        b = [(xa, 300, xa + 40, 420), (xb, 300, xb + 40, 420), (900, 200, 940, 320)]
        boxes.append(b)
        dets.append(np.array([[*b[0], 0.90, 0], [*b[1], 0.88, 0], [*b[2], 0.86, 0]], np.float64))
    return dets, boxes


def _records(dets):
    recs = np.zeros(len(dets), lib.RESULT_DTYPE)
    for i, d in enumerate(dets):
        recs[i]["n_det"] = len(d)
        for k in range(len(d)):
            e = recs[i]["det"][k]
            e["x1"], e["y1"], e["x2"], e["y2"], e["conf"], e["cls"] = d[k]
            e["id"] = k; e["reported"] = 1
            e["bx1"], e["by1"], e["bx2"], e["by2"] = [int(v) for v in d[k][:4]]
            e["foot_x"], e["foot_y"] = int((int(d[k][0]) + int(d[k][2])) / 2), int(d[k][3])
    return recs


def test_track_ids_with_appearance_through_a_crossing():
    dets, boxes = _crossing_clip()
    frames = _render(boxes, COLORS)
    sd = osnet.make_osnet_state_dict(0)
    from eagle_amd.coordinate_model import CoordinateModel
    cm = CoordinateModel(batch=1, tracker=True, reid=True, reid_state_dict=sd)
    recs = _records(dets)
    crops, det, count = cm.reid_inputs(recs)
    assert count.tolist() == [3] * len(dets)
    d = cm.handle.upload(frames)
    feats = cm.handle.reid_features(d, len(frames), crops)
    cm.handle.free(d)
    cm.handle.track_open()
    cm.handle.track_frames_reid(recs, feats, det, count)
    # without appearance on the same detections (the round-2 behaviour, unchanged)
    recs0 = _records(dets)
    cm.handle.track_open()
    cm.handle.track_frames(recs0)
    cm.handle.close()
    # oracle: same tracker restatement, embeddings from the torch OSNet
    tr, tr0 = BotSortLite(), BotSortLite()
    ids_gpu, ids_ora, ids_gpu0, ids_ora0 = [], [], [], []
    for i, dd in enumerate(dets):
        fo = reid.features(sd, frames[i], dd[:, :4])
        out = tr.update(dd, feats={k: fo[k] for k in range(len(dd))})
        out0 = tr0.update(dd)
        ids_ora.append({int(r[7]): int(r[4]) for r in out}); ids_ora0.append({int(r[7]): int(r[4]) for r in out0})
        ids_gpu.append({k: int(recs[i]["det"][k]["id"]) for k in range(len(dd)) if recs[i]["det"][k]["reported"] and out.size})
        ids_gpu0.append({k: int(recs0[i]["det"][k]["id"]) for k in range(len(dd)) if recs0[i]["det"][k]["reported"] and out0.size})
    for i in range(len(dets)):
        if ids_ora[i]:
            assert ids_gpu[i] == ids_ora[i], (i, ids_gpu[i], ids_ora[i])
        if ids_ora0[i]:
            assert ids_gpu0[i] == ids_ora0[i], (i, ids_gpu0[i], ids_ora0[i])
    # appearance takes part in the association: while the two boxes coincide (frames 15..19) IoU cannot order the pairs and the
    # embedding distances decide — the id maps of the two modes differ there
    differ = [i for i in range(len(dets)) if ids_ora[i] != ids_ora0[i]]
    print("frames on which appearance changes the association:", differ)
    assert differ and all(14 <= i <= 20 for i in differ[:1]), differ


def test_coordinate_model_with_reid_runs_and_matches_its_own_parts():
    """CoordinateModel(tracker=True, reid=True): the reference's configuration (BotSort with appearance) end to end on synthetic frames —
    the ids equal those of the same records tracked through the explicit calls (reid_inputs -> eagle_reid_features -> eagle_track_frames_reid)."""
    from eagle_amd import synth
    from eagle_amd.coordinate_model import CoordinateModel
    frames = np.stack([synth.frame(0, t) for t in range(5)])
    cm = CoordinateModel(batch=2, tracker=True, reid=True, detector_conf=0.2)
    res = cm.get_coordinates(frames, fps=1)
    recs = cm.process_records(frames)
    crops, det, count = cm.reid_inputs(recs)
    d = cm.handle.upload(frames)
    feats = cm.handle.reid_features(d, len(frames), crops)
    cm.handle.free(d)
    cm.reset_tracker(); cm.handle.track_open()
    cm.handle.track_frames_reid(recs, feats, det, count)
    from eagle_amd import records
    again = {i: records.to_reference_dict(r, i, 1) for i, r in enumerate(recs)}
    cm.handle.close()
    assert len(crops) > 0 and feats.shape == (len(crops), 512) and np.isfinite(feats).all()
    for i in res:
        assert res[i]["Coordinates"] == again[i]["Coordinates"], i

"""-m gpu: the whole per-frame path through the C ABI against the CPU oracle on the same seeded frames/weights.

fp32 handle: every integer (heat-map indices, keypoint pixels, NMS order/IDs, boxes, pitch ints) and H must be
IDENTICAL to the oracle (scores/floats bit-equal too: same fmaf chains, same exp polynomial).
fp16 handle (the fast path): heat-map maxima compared with near-tie handling, boxes by IoU, H by reprojection."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def frames():
    from eagle_amd import synth
    return np.stack([synth.frame(0, 0), synth.frame(0, 37), synth.noise_frame(1), synth.frame(0, 12), synth.frame(2, 5)])


@pytest.fixture(scope="module")
def oracle_steps(state_dicts, frames):
    from oracle import pipeline
    hs, ys = state_dicts
    m = pipeline.OracleModel(hs, ys, backend="c")
    return [m.step(f, i) for i, f in enumerate(frames)]


def _canon(d):
    """dict -> comparable plain structure (tuples -> lists, numpy scalars -> python)."""
    if isinstance(d, dict):
        return {str(k): _canon(v) for k, v in d.items() if not str(k).startswith("_")}
    if isinstance(d, (list, tuple)):
        return [_canon(v) for v in d]
    if isinstance(d, (np.integer,)):
        return int(d)
    if isinstance(d, (np.floating,)):
        return float(d)
    return d


def test_f32_path_identical_to_oracle(state_dicts, frames, oracle_steps):
    from eagle_amd import records
    from eagle_amd.coordinate_model import CoordinateModel
    hs, ys = state_dicts
    cm = CoordinateModel(precision="f32", batch=2, hrnet_state_dict=hs, detector_state_dict=ys)
    recs = cm.process_records(frames)          # 3 frames with batch 2: exercises the ragged last batch
    for i, (rec, (oref, aux)) in enumerate(zip(recs, oracle_steps)):
        assert np.array_equal(rec["hm_idx"], aux["hm_idx"]), f"frame {i}: heat-map argmax differs"
        assert np.array_equal(rec["hm_score"], aux["hm_score"]), f"frame {i}: heat-map scores differ"
        n = int(rec["n_det"])
        assert n == len(aux["dets"]), f"frame {i}: {n} detections vs {len(aux['dets'])}"
        got = np.stack([rec["det"][k][:n] for k in ("x1", "y1", "x2", "y2", "conf")], 1)
        assert np.array_equal(got, aux["dets"][:, :5]), f"frame {i}: boxes/conf differ"
        assert np.array_equal(rec["det"]["cls"][:n], aux["dets"][:, 5].astype(np.int32))
        if aux["H"] is None:
            assert not rec["H_valid"]
        else:
            assert rec["H_valid"] and np.array_equal(rec["H"].reshape(3, 3), aux["H"]), f"frame {i}: H differs"
        assert _canon(records.to_reference_dict(rec, i)) == _canon(oref), f"frame {i}: record differs"
    cm.handle.close()


def test_f16_path_within_tolerance(state_dicts, frames, oracle_steps):
    from eagle_amd.coordinate_model import CoordinateModel
    from oracle import prims as P
    hs, ys = state_dicts
    cm = CoordinateModel(precision="f16", batch=3, hrnet_state_dict=hs, detector_state_dict=ys)
    recs = cm.process_records(frames)
    exact = total = 0
    for i, (rec, (oref, aux)) in enumerate(zip(recs, oracle_steps)):
        sig = P.sigmoid(aux["logits"][0])                  # fp32 oracle heat-maps [135,240,57]
        flat = sig.reshape(-1, 57)
        for c in range(57):
            total += 1
            gi = int(rec["hm_idx"][c])
            if gi == int(aux["hm_idx"][c]):
                exact += 1
            # near-tie rule: the fp16 path's maximum must be (almost) as high as the fp32 maximum
            assert flat[gi, c] >= aux["hm_score"][c] - 0.02, f"frame {i} ch {c}: fp16 argmax is not a near-maximum"
            assert abs(float(rec["hm_score"][c]) - float(aux["hm_score"][c])) < 0.02
        assert abs(int(rec["n_candidates"]) - int((aux["rows"][:, 4:].max(1) > 0.15).sum())) <= 0.1 * max(50, rec["n_candidates"])
    assert exact >= 0.85 * total, f"only {exact}/{total} heat-map maxima identical between fp16 path and fp32 oracle"
    cm.handle.close()


def test_batch_invariance(state_dicts, frames):
    """Results must not depend on the batch size or on a frame's position in the batch (fp16 path)."""
    from eagle_amd.coordinate_model import CoordinateModel
    hs, ys = state_dicts
    a = CoordinateModel(precision="f16", batch=1, hrnet_state_dict=hs, detector_state_dict=ys)
    ra = a.process_records(frames)
    a.handle.close()
    b = CoordinateModel(precision="f16", batch=4, hrnet_state_dict=hs, detector_state_dict=ys, use_graph=True)
    rb = b.process_records(frames[::-1])[::-1]
    rb2 = b.process_records(frames[::-1])[::-1]          # graph replay
    b.handle.close()
    assert ra.tobytes() == rb.tobytes() == rb2.tobytes()


def test_full_size_clip_properties(state_dicts):
    """BASELINE.json's configs[1] at full size (1000 frames of 1280x720, device batch 50, the bench's clip: 20 distinct frames tiled): every copy of a
    frame yields byte-identical records wherever it sits in the clip and in its batch; a different device batch (37: ragged last batch) yields the
    same 1000 records; and the host-fed path (pageable memory through the pinned ring) equals the resident path."""
    from eagle_amd import synth
    from eagle_amd.coordinate_model import CoordinateModel
    hs, ys = state_dicts
    base = np.stack([synth.frame(0, t) for t in range(20)])
    clip = np.ascontiguousarray(np.tile(base, (50, 1, 1, 1)))
    a = CoordinateModel(precision="f16", batch=50, hrnet_state_dict=hs, detector_state_dict=ys)
    ra = a.process_records(clip)                                  # host frames: eagle_process_frames
    d = a.handle.upload(clip)
    rd = np.zeros(len(clip), ra.dtype)
    a.handle.process_device(d, len(clip), rd)                     # resident frames: eagle_process_device_frames
    a.handle.free(d); a.handle.close()
    assert ra.tobytes() == rd.tobytes()
    for t in range(20):
        first = ra[t].tobytes()
        assert all(ra[k].tobytes() == first for k in range(t, 1000, 20)), t
    b = CoordinateModel(precision="f16", batch=37, hrnet_state_dict=hs, detector_state_dict=ys)
    rb = b.process_records(clip)
    b.handle.close()
    assert ra.tobytes() == rb.tobytes()


def test_rccl_gather_single_rank(state_dicts):
    """eagle_comm_id / eagle_comm_init / eagle_gather through the real RCCL library (world size 1 on this box)."""
    from eagle_amd import lib
    h = lib.Handle(batch=1)
    uid = lib.comm_unique_id()
    assert len(uid) == 128 and any(uid)
    h.comm_init(0, 1, uid)
    rec = np.zeros(5, lib.RESULT_DTYPE)
    rec["n_det"] = np.arange(5)
    rec["H"][:, 3] = 2.5
    out = h.gather(rec, 1)
    assert out.tobytes() == rec.tobytes()
    h.close()


def test_cfg3_large_detector_1080p_identical_to_oracle():
    """BASELINE.json configs[2]: 1920x1080 frames, YOLOv8-l at imgsz 960 (544x960 letterbox) + HRNet-W48.
    fp32 handle vs the exact-order oracle on one frame; algorithmic FLOP/frame must be 544.3 G."""
    from eagle_amd import records, synth, weights
    from eagle_amd.coordinate_model import CoordinateModel
    from oracle import pipeline
    hs, yl = weights.make_hrnet_state_dict(0), weights.make_yolo_state_dict("l", 0)
    frame = synth.frame(0, 3, 1080, 1920)
    cm = CoordinateModel(precision="f32", batch=1, frame_hw=(1080, 1920), detector="l", det_imgsz=960,
                         hrnet_state_dict=hs, detector_state_dict=yl)
    rec = cm.process_records(frame[None])[0]
    cm.handle.set_profiling(1)
    cm.process_records(frame[None])
    t = cm.handle.timings()
    assert abs(t.conv_flop / 1e9 - 544.3) < 0.2, t.conv_flop
    cm.handle.close()
    ora = pipeline.OracleModel(hs, yl, variant="l", imgsz=960, backend="c")
    oref, aux = ora.step(frame, 0)
    assert np.array_equal(rec["hm_idx"], aux["hm_idx"])
    n = int(rec["n_det"])
    assert n == len(aux["dets"])
    assert np.array_equal(np.stack([rec["det"][k][:n] for k in ("x1", "y1", "x2", "y2", "conf")], 1), aux["dets"][:, :5])
    assert np.array_equal(rec["det"]["cls"][:n], aux["dets"][:, 5].astype(np.int32))
    assert _canon(records.to_reference_dict(rec, 0)) == _canon(oref)


def test_f16_path_vs_f16_emulating_oracle(state_dicts, frames):
    """fp16 kernels vs the oracle run with fp16 STORAGE emulation (tensors/weights rounded to binary16 at the points
    where the kernels store): the only remaining difference is the MFMA's internal summation order, so agreement must
    be much tighter than against the fp32 oracle."""
    from eagle_amd.coordinate_model import CoordinateModel
    from oracle import host, nets, prims as P
    hs, ys = state_dicts
    cm = CoordinateModel(precision="f16", batch=1, hrnet_state_dict=hs, detector_state_dict=ys)
    rec = cm.process_records(frames[:1])[0]
    cm.handle.close()
    lg = nets.hrnet_logits(hs, host.preprocess_keypoints(frames[0]), backend="c", f16=True)
    idx, score = P.heatmap_argmax(lg[0], 57)
    same = int((rec["hm_idx"] == idx).sum())
    assert same >= 54, f"only {same}/57 heat-map maxima identical to the fp16-emulating oracle"
    assert np.abs(rec["hm_score"] - score).max() < 5e-3
    sig = P.sigmoid(lg[0]).reshape(-1, 57)
    for c in range(57):
        assert sig[int(rec["hm_idx"][c]), c] >= score[c] - 5e-3
    x, g = host.preprocess_detector(frames[0], 640)
    rows = nets.yolo_decode(nets.yolo_heads(ys, x, "n", backend="c", f16=True))
    dets = host.nms_and_scale(rows, 720, 1280, g["out_h"], g["out_w"])
    n = int(rec["n_det"])
    assert abs(n - len(dets)) <= max(8, 0.05 * len(dets))      # candidates sitting on the 0.15 floor / 0.7 IoU may flip
    # order-insensitive: every confident oracle detection must have a GPU detection of the same class with IoU > 0.9
    # (confidences that differ in the 4th digit permute the descending-confidence order)
    g = np.stack([rec["det"][k][:n] for k in ("x1", "y1", "x2", "y2")], 1)
    gcls = rec["det"]["cls"][:n]

    def iou(a, b):
        x1 = np.maximum(a[0], b[:, 0]); y1 = np.maximum(a[1], b[:, 1]); x2 = np.minimum(a[2], b[:, 2]); y2 = np.minimum(a[3], b[:, 3])
        inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
        return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) - inter + 1e-9)
    conf = dets[dets[:, 4] >= 0.25]
    hit = [bool(((iou(d[:4], g) > 0.9) & (gcls == int(d[5]))).any()) for d in conf]
    assert len(conf) > 10 and np.mean(hit) > 0.95, f"{np.mean(hit):.2f} of {len(conf)} confident oracle detections found by the fp16 path"


# ---- record-level parity of the fp16 (benchmarked) family ------------------------------------------------------------------------------
def _peaked_state_dict(hs):
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "peaked_head.npz"))
    hs2 = dict(hs)
    hs2["unnormalized_model.1.weight"] = g["weight"]; hs2["unnormalized_model.1.bias"] = g["bias"]
    return hs2, g


def _iou(a, b):
    x1 = np.maximum(a[0], b[:, 0]); y1 = np.maximum(a[1], b[:, 1]); x2 = np.minimum(a[2], b[:, 2]); y2 = np.minimum(a[3], b[:, 3])
    inter = np.clip(x2 - x1, 0, None) * np.clip(y2 - y1, 0, None)
    return inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1]) - inter + 1e-9)


def _f16_record_parity(rec, oref, aux, tag, frame_hw, conf_tol=2.5e-3):
    """One frame: the fp16 family's record against the fp16-STORAGE-emulating oracle (same rounding points, exact-order fp32 sums).
    Returns counters for the summary line."""
    from eagle_amd.pitch import INTERSECTION_TO_PITCH_POINTS
    from oracle import prims as P
    sig = P.sigmoid(aux["logits"][0]).reshape(-1, 57)
    srt = np.sort(sig, 0)
    margin = srt[-1] - srt[-2]
    peaked = np.nonzero(margin > 0.05)[0]
    flat = np.nonzero(margin == 0)[0]                   # constant maps (landmark not in view): first index on both sides
    # (1) heat-map maxima: identical wherever the oracle's own top-1 / top-2 margin exceeds 0.05
    bad = [int(c) for c in list(peaked) + list(flat) if int(rec["hm_idx"][c]) != int(aux["hm_idx"][c])]
    assert not bad, f"{tag}: heat-map maxima differ on channels with a clear maximum: {bad}"
    # (the matched-filter head multiplies the backbone's fp16 noise by ALPHA / |patch|^2: the peak VALUE moves by up to ~5e-3 — measured —
    #  while its position, which is what the reference consumes, does not)
    assert np.abs(rec["hm_score"][peaked] - aux["hm_score"][peaked]).max(initial=0) < 2e-2
    # (2) key-point pixels (cm.py:500-518) identical
    kp = {INTERSECTION_TO_PITCH_POINTS[int(k["label"])]: (int(k["x"]), int(k["y"])) for k in rec["kp"][: int(rec["n_kp"])] if not k["synthesized"]}
    same_kp = kp == {k: (int(v[0]), int(v[1])) for k, v in aux["kp_detected"].items()}
    # (3) with identical key-points the geometry must be IDENTICAL (the geometry kernel does not depend on the family): synthesis, H, inliers
    h_checked = False
    if same_kp:
        allkp = {INTERSECTION_TO_PITCH_POINTS[int(k["label"])]: (int(k["x"]), int(k["y"])) for k in rec["kp"][: int(rec["n_kp"])]}
        assert allkp == {k: (int(v[0]), int(v[1])) for k, v in aux["kp_synth"].items()}, f"{tag}: synthesised key-points differ"
        assert bool(rec["H_valid"]) == (aux["H"] is not None), f"{tag}: H validity differs"
        if aux["H"] is not None:
            Hg, Ho = rec["H"].reshape(3, 3), aux["H"]
            assert np.abs(Hg - Ho).max() <= 1e-3 * np.abs(Ho).max() and np.allclose(Hg, Ho, rtol=1e-3, atol=1e-9), f"{tag}: H differs beyond 1e-3"
            h_checked = True
    # (4) detections: every oracle detection has a GPU detection of the same class at IoU > 0.9 (95 %); integer boxes within 1 px
    n = int(rec["n_det"]); dets = aux["dets"]
    assert abs(n - len(dets)) <= max(3, 0.05 * len(dets)), f"{tag}: {n} detections vs {len(dets)}"
    g = np.stack([rec["det"][k][:n] for k in ("x1", "y1", "x2", "y2")], 1) if n else np.zeros((0, 4), np.float32)
    gcls = rec["det"]["cls"][:n]
    fh, fw = frame_hw
    match, nonident, checked_pitch, maxoff = [], 0, 0, 0
    for i, d in enumerate(dets):
        if n == 0:
            match.append(-1); continue
        io = _iou(d[:4], g) * (gcls == int(d[5]))
        j = int(io.argmax())
        match.append(j if io[j] > 0.9 else -1)
    found = [m for m in match if m >= 0]
    assert len(dets) == 0 or len(found) >= 0.95 * len(dets), f"{tag}: only {len(found)} of {len(dets)} oracle detections found"
    for i, j in enumerate(match):
        if j < 0:
            continue
        d = dets[i]
        bi = np.array([int(d[0]), int(d[1]), int(d[2]), int(d[3])])              # astype(int) truncation (cm.py:600), before clipping
        is_person = int(d[5]) in (0, 1)
        if is_person:
            bi = np.array([min(max(bi[0], 0), fw - 1), min(max(bi[1], 0), fh - 1), min(max(bi[2], 0), fw - 1), min(max(bi[3], 0), fh - 1)])
        gi = np.array([int(rec["det"][k][j]) for k in ("bx1", "by1", "bx2", "by2")])
        # one pixel of the NETWORK input is 1/gain = 2 frame pixels in both configurations (scale_boxes divides by 0.5)
        assert np.abs(gi - bi).max() <= 2, f"{tag}: integer box of detection {i} off by more than a network-input pixel: {gi} vs {bi}"
        nonident += int((gi != bi).any()); maxoff = max(maxoff, int(np.abs(gi - bi).max()))
    # confidences: the fp16 MFMA's summation order moves a confidence by up to 1.4e-3 through yolov8n's 63 convolutions and 4.4e-3
    # through yolov8l's 103 on these random-weight detectors (measured: the value is returned and printed; bounded by conf_tol) ...
    dconf = max([abs(float(rec["det"]["conf"][j]) - float(dets[i][4])) for i, j in enumerate(match) if j >= 0], default=0.0)
    assert dconf < conf_tol, f"{tag}: a confidence moved by {dconf}"
    # ... so the ID order (descending confidence, cm.py:598-616) is kept wherever two confidences are further apart than both can move
    for i in range(len(dets) - 1):
        if match[i] >= 0 and match[i + 1] >= 0 and dets[i][4] - dets[i + 1][4] > 2 * conf_tol:
            assert match[i] < match[i + 1], f"{tag}: detection order differs where confidences are {dets[i][4]} / {dets[i + 1][4]}"
    # (5) pitch coordinates of the reported objects: identical ints and floats within 1e-3 wherever foot point and H are identical
    if h_checked and np.array_equal(rec["H"].reshape(3, 3), aux["H"]):
        oobj = oref["Coordinates"]
        for cname in ("Player", "Goalkeeper"):
            for oid, o in oobj.get(cname, {}).items():
                j = match[int(oid)] if int(oid) < len(match) else -1
                if j < 0:
                    continue
                gd = rec["det"][j]
                ofoot = o.get("Image_Bottom_center") or [int((o["BBox"][0] + o["BBox"][2]) / 2), o["BBox"][3]]
                if [int(gd["foot_x"]), int(gd["foot_y"])] != [int(ofoot[0]), int(ofoot[1])]:
                    continue
                tc = o["Transformed_Coordinates"]
                assert bool(gd["in_bounds"]) == (tc is not None), f"{tag}: in-bounds flag differs for object {oid}"
                if tc is not None:
                    assert [int(gd["pitch_x"]), int(gd["pitch_y"])] == [int(tc[0]), int(tc[1])], f"{tag}: pitch integers differ for object {oid}"
                checked_pitch += 1
    return dict(peaked=len(peaked), max_conf_dev=round(dconf, 6), same_kp=same_kp, h_checked=h_checked, dets=len(dets), found=len(found), nonident=nonident, max_box_px=maxoff, pitch=checked_pitch)


def test_f16_family_record_parity_cfg2(state_dicts):
    """The benchmarked family on cfg 2 (1280x720, yolov8n@640 + HRNet-W48) with the peaked-heat-map head of
    tests/golden/make_peaked_head.py: heat-map maxima identical on every channel with a clear maximum (all 57 on the design frame),
    key-point pixels, synthesised points, H and pitch coordinates identical to the fp16-emulating oracle, integer boxes within a pixel."""
    from eagle_amd import synth
    from eagle_amd.coordinate_model import CoordinateModel
    from oracle import pipeline
    hs, ys = state_dicts
    hs2, g = _peaked_state_dict(hs)
    frames = np.stack([synth.frame(*g["design"]), synth.frame(0, 4), synth.frame(0, 9), synth.frame(0, 1), synth.frame(0, 6), synth.frame(0, 14)])
    cm = CoordinateModel(precision="f16", batch=2, hrnet_state_dict=hs2, detector_state_dict=ys)
    recs = cm.process_records(frames)
    cm.handle.close()
    ora = pipeline.OracleModel(hs2, ys, backend="c", f16=True)
    tot = []
    for i, f in enumerate(frames):
        oref, aux = ora.step(f, i)
        tot.append(_f16_record_parity(recs[i], oref, aux, f"cfg2 frame {i}", (720, 1280)))
    print("fp16 record parity cfg2:", tot)
    assert tot[0]["peaked"] >= 20 and tot[0]["same_kp"] and tot[0]["h_checked"], tot[0]       # the design frame: every placed landmark, H solved and identical
    assert sum(t["nonident"] for t in tot) <= 0.25 * max(1, sum(t["found"] for t in tot)), tot
    # the fp32 (exact) family on the same peaked weights equals the fp32 oracle bit for bit, like on the random head
    cm32 = CoordinateModel(precision="f32", batch=1, hrnet_state_dict=hs2, detector_state_dict=ys)
    r32 = cm32.process_records(frames[:1])[0]
    cm32.handle.close()
    _, aux32 = pipeline.OracleModel(hs2, ys, backend="c").step(frames[0], 0)
    assert np.array_equal(r32["hm_idx"], aux32["hm_idx"]) and np.array_equal(r32["hm_score"], aux32["hm_score"])
    assert r32["H_valid"] and np.array_equal(r32["H"].reshape(3, 3), aux32["H"])


def test_f16_family_record_parity_cfg3():
    """The same on cfg 3 (1920x1080, yolov8l@960)."""
    from eagle_amd import synth, weights
    from eagle_amd.coordinate_model import CoordinateModel
    from oracle import pipeline
    hs, yl = weights.make_hrnet_state_dict(0), weights.make_yolo_state_dict("l", 0)
    hs2, g = _peaked_state_dict(hs)
    frame = synth.frame(int(g["design"][0]), int(g["design"][1]), 1080, 1920)
    cm = CoordinateModel(precision="f16", batch=1, frame_hw=(1080, 1920), detector="l", det_imgsz=960, hrnet_state_dict=hs2, detector_state_dict=yl)
    rec = cm.process_records(frame[None])[0]
    cm.handle.close()
    oref, aux = pipeline.OracleModel(hs2, yl, variant="l", imgsz=960, backend="c", f16=True).step(frame, 0)
    t = _f16_record_parity(rec, oref, aux, "cfg3", (1080, 1920), conf_tol=1e-2)
    print("fp16 record parity cfg3:", t)
    assert t["nonident"] <= 0.25 * max(1, t["found"]), t


# ---- record-level parity of the split-precision family (EAGLE_PREC_F32S) against the fp32 oracle --------------------------------------------
def _near_int(v, tol=1e-3):
    return abs(v - round(v)) <= tol


def _f32s_record_parity(rec, oref, aux, tag, frame_hw, conf_tol=5e-6, box_tol=2e-3, score_tol=5e-6):
    """One frame of the split family against the fp32 oracle (OracleModel(backend="c"), the one pinned to the reference) — NOT an emulation
    of the family's own rounding.  north_star's contract: integers identical, floats within 1e-3 relative; what is asserted here is much
    tighter: every integer field identical, floats to a few 1e-6.  The only admitted integer differences are the two SURVEY §0 names
    — a float that sits within 1e-3 of an integer before truncation, and an arg-max between two heat-map values closer than 2e-6 — and each
    occurrence is counted and returned so that the summary line shows them.
    Float tolerances are set from the spread between the oracle's OWN two fp32 backends (exact fmaf chain vs torch / oneDNN, i.e. two legitimate
    fp32 summation orders; tests/golden/fp32_order_noise.py): cfg 2 — confidences 2.4e-6, heat-map scores 2.4e-7, logits 8.8e-7 relative, and
    one swapped pair of detection ids; cfg 3 (yolov8l, 103 convolutions deep) — confidences 2.0e-5 .. 3.0e-5 over the three test frames, boxes
    5.3e-3 px.  conf_tol / box_tol are 2x those.  The family's measured deviations (printed) equal the fp32 spread on cfg 2 (2.5e-6) and reach
    1.8x it on the deepest network (4.3e-5 on one cfg-3 frame: 103 layers of 22-bit tensor storage against fp32's 24)."""
    from eagle_amd import records
    from eagle_amd.pitch import INTERSECTION_TO_PITCH_POINTS
    from oracle import prims as P
    out = dict(hm_tie=0, near_int_box=0, conf_tie=0, max_score_dev=0.0, max_conf_dev=0.0, max_box_dev=0.0, max_pitch_dev=0.0, h_identical=None, dets=0, kp_same=True)
    # (1) heat-map maxima
    sig = P.sigmoid(aux["logits"][0]).reshape(-1, 57)
    for c in range(57):
        gi, oi = int(rec["hm_idx"][c]), int(aux["hm_idx"][c])
        if gi != oi:
            assert sig[gi, c] >= aux["hm_score"][c] * (1 - 2e-6), f"{tag} ch {c}: arg-max {gi} vs {oi} is not a near-tie ({sig[gi, c]} vs {aux['hm_score'][c]})"
            out["hm_tie"] += 1
    sdev = float(np.abs(rec["hm_score"].astype(np.float64) - aux["hm_score"]).max())
    assert sdev <= score_tol, f"{tag}: heat-map score moved by {sdev}"
    out["max_score_dev"] = sdev
    # (2) key-points: pixels, labels, synthesised points
    kp = {INTERSECTION_TO_PITCH_POINTS[int(k["label"])]: (int(k["x"]), int(k["y"])) for k in rec["kp"][: int(rec["n_kp"])] if not k["synthesized"]}
    okp = {k: (int(v[0]), int(v[1])) for k, v in aux["kp_detected"].items()}
    if out["hm_tie"] == 0:
        # a score within 5e-6 of the 0.3 threshold could flip membership: none of the synthetic frames has one (asserted)
        assert kp == okp, f"{tag}: key-points differ: {set(kp.items()) ^ set(okp.items())}"
    out["kp_same"] = kp == okp
    if out["kp_same"]:
        allkp = {INTERSECTION_TO_PITCH_POINTS[int(k["label"])]: (int(k["x"]), int(k["y"])) for k in rec["kp"][: int(rec["n_kp"])]}
        assert allkp == {k: (int(v[0]), int(v[1])) for k, v in aux["kp_synth"].items()}, f"{tag}: synthesised key-points differ"
        assert bool(rec["H_valid"]) == (aux["H"] is not None), f"{tag}: H validity differs"
        if aux["H"] is not None:      # same integer key-points -> the geometry kernel (family-independent, fp64) must give the same bits
            assert np.array_equal(rec["H"].reshape(3, 3), aux["H"]), f"{tag}: H differs: {np.abs(rec['H'].reshape(3, 3) - aux['H']).max()}"
            out["h_identical"] = True
    # (3) detections, index by index (the index IS the id, cm.py:598-616)
    dets = aux["dets"]
    n = int(rec["n_det"])
    out["dets"] = len(dets)
    assert n == len(dets), f"{tag}: {n} detections vs {len(dets)}"
    fh, fw = frame_hw
    for i, d in enumerate(dets):
        # detection i of the oracle is detection i of the GPU — or, where two neighbouring confidences are closer than the family's own
        # deviation (a few 1e-7), its neighbour: that swap of ids is the "confidence near-tie" the summary counts
        def fits(j):
            gj = rec["det"][j]
            return (int(gj["cls"]) == int(d[5]) and abs(float(gj["conf"]) - float(d[4])) <= conf_tol and
                    float(np.abs(np.array([float(gj[k]) for k in ("x1", "y1", "x2", "y2")]) - d[:4]).max()) <= box_tol)
        js = [j for j in (i, i - 1, i + 1) if 0 <= j < n and fits(j)]
        assert js, f"{tag}: detection {i} (cls {d[5]}, conf {d[4]}, box {d[:4]}) has no counterpart at ids {i - 1}..{i + 1}: GPU has {rec['det'][i]}"
        if js[0] != i:
            assert abs(float(dets[js[0]][4]) - float(d[4])) <= 2 * conf_tol, f"{tag}: detections {i} / {js[0]} swapped without a confidence near-tie"
            out["conf_tie"] += 1
            continue
        g = rec["det"][i]
        out["max_conf_dev"] = max(out["max_conf_dev"], abs(float(g["conf"]) - float(d[4])))
        out["max_box_dev"] = max(out["max_box_dev"], float(np.abs(np.array([float(g[k]) for k in ("x1", "y1", "x2", "y2")]) - d[:4]).max()))
        bi = np.array([int(d[0]), int(d[1]), int(d[2]), int(d[3])])              # astype(int) truncation (cm.py:600), before clipping
        if int(d[5]) in (0, 1):
            bi = np.array([min(max(bi[0], 0), fw - 1), min(max(bi[1], 0), fh - 1), min(max(bi[2], 0), fw - 1), min(max(bi[3], 0), fh - 1)])
        gi = np.array([int(g[k]) for k in ("bx1", "by1", "bx2", "by2")])
        for a in range(4):
            if gi[a] != bi[a]:
                assert abs(int(gi[a]) - int(bi[a])) == 1 and _near_int(float(d[a])), f"{tag}: integer box of detection {i} differs away from an integer boundary: {gi} vs {bi} ({d[:4]})"
                out["near_int_box"] += 1
    # (4) the reference-schema record: identical keys / ints / None-ness; floats (confidences, pitch floats) to 1e-5 relative
    if out["hm_tie"] == 0 and out["near_int_box"] == 0 and out["conf_tie"] == 0:
        got, ref = _canon_keep(records.to_reference_dict(rec, 0)), _canon_keep(oref)
        _assert_same(got, ref, tag, ftol=max(1e-5, conf_tol))
        for cname in ("Player", "Goalkeeper", "Ball"):
            for oid, o in oref["Coordinates"].get(cname, {}).items():
                if o.get("_pitch_float") is None:
                    continue
                j = int(oid) if cname != "Ball" else [k for k in range(n) if int(rec["det"][k]["cls"]) == 2][int(oid)]
                gd = rec["det"][j]
                pdev = max(abs(float(gd["pitch_xf"]) - o["_pitch_float"][0]), abs(float(gd["pitch_yf"]) - o["_pitch_float"][1]))
                assert pdev <= 1e-6 * max(1.0, abs(o["_pitch_float"][0]), abs(o["_pitch_float"][1])), f"{tag}: pitch float of {cname} {oid} moved by {pdev}"
                out["max_pitch_dev"] = max(out["max_pitch_dev"], pdev)
    return out


def _canon_keep(d):
    if isinstance(d, dict):
        return {str(k): _canon_keep(v) for k, v in d.items() if not str(k).startswith("_")}
    if isinstance(d, (list, tuple)):
        return [_canon_keep(v) for v in d]
    if isinstance(d, np.integer):
        return int(d)
    if isinstance(d, np.floating):
        return float(d)
    return d


def _assert_same(a, b, tag, path="", ftol=1e-5):
    if isinstance(b, dict):
        assert isinstance(a, dict) and a.keys() == b.keys(), f"{tag}{path}: keys {sorted(a) if isinstance(a, dict) else a} vs {sorted(b)}"
        for k in b:
            _assert_same(a[k], b[k], tag, f"{path}/{k}", ftol)
    elif isinstance(b, list):
        assert isinstance(a, list) and len(a) == len(b), f"{tag}{path}: {a} vs {b}"
        for k, (x, y) in enumerate(zip(a, b)):
            _assert_same(x, y, tag, f"{path}[{k}]", ftol)
    elif isinstance(b, float) and not isinstance(b, bool):
        assert isinstance(a, float) and abs(a - b) <= ftol * max(1.0, abs(b)), f"{tag}{path}: {a} vs {b}"
    else:
        assert type(a) is type(b) and a == b, f"{tag}{path}: {a!r} vs {b!r}"


def test_f32s_family_equals_fp32_oracle_random_head(state_dicts, frames, oracle_steps):
    """The split-precision family against the fp32 oracle on the five cfg-2 frames of the exact family's own test (seeded random weights:
    noise-like heat-maps, H mostly unsolvable) — same fixtures, same oracle steps as test_f32_path_identical_to_oracle."""
    from eagle_amd.coordinate_model import CoordinateModel
    hs, ys = state_dicts
    cm = CoordinateModel(precision="f32s", detector_precision="f32s", batch=2, hrnet_state_dict=hs, detector_state_dict=ys)
    recs = cm.process_records(frames)
    cm.handle.close()
    tot = [_f32s_record_parity(recs[i], oref, aux, f"f32s random head frame {i}", (720, 1280)) for i, (oref, aux) in enumerate(oracle_steps)]
    print("f32s parity (random head):", tot)
    # The random-weight detector reports 240 - 300 boxes per frame with confidences ~1e-3 apart, so a few neighbouring ids swap under ANY
    # change of fp32 summation order (the oracle's own two backends swap a pair on frame 0): each swap is verified as a near-tie above and
    # counted; test_f32s_ids_identical_with_a_sparse_detector is the same comparison with a realistic number of detections.
    assert sum(t["hm_tie"] for t in tot) <= 2 and sum(t["near_int_box"] for t in tot) <= 2, tot
    assert sum(t["conf_tie"] for t in tot) <= 0.02 * sum(t["dets"] for t in tot), tot
    assert sum(t["dets"] for t in tot) > 20


def test_f32s_family_equals_fp32_oracle_peaked_head_cfg2(state_dicts):
    """The same with the peaked-heat-map head (geometrically consistent key-points, H solvable): integer key-points identical -> H and the
    pitch floats bit-identical / within 1e-6."""
    from eagle_amd import synth
    from eagle_amd.coordinate_model import CoordinateModel
    from oracle import pipeline
    hs, ys = state_dicts
    hs2, g = _peaked_state_dict(hs)
    frames = np.stack([synth.frame(*g["design"]), synth.frame(0, 4), synth.frame(0, 9), synth.frame(0, 1), synth.frame(0, 6), synth.frame(0, 14)])
    cm = CoordinateModel(precision="f32s", detector_precision="f32s", batch=4, hrnet_state_dict=hs2, detector_state_dict=ys)
    recs = cm.process_records(frames)
    cm.handle.close()
    ora = pipeline.OracleModel(hs2, ys, backend="c")
    tot = []
    for i, f in enumerate(frames):
        oref, aux = ora.step(f, i)
        # the matched-filter head multiplies backbone deviations by ALPHA / |patch|^2 (make_peaked_head.py): peak VALUES move ~20x more than on
        # the random head, peak positions do not move
        tot.append(_f32s_record_parity(recs[i], oref, aux, f"f32s peaked cfg2 frame {i}", (720, 1280), score_tol=3e-5))
    print("f32s parity (peaked head, cfg2):", tot)
    assert sum(bool(t["h_identical"]) for t in tot) >= 5, tot
    assert sum(t["hm_tie"] + t["near_int_box"] for t in tot) <= 2, tot
    assert sum(t["conf_tie"] for t in tot) <= 0.02 * sum(t["dets"] for t in tot), tot


def test_f32s_family_equals_fp32_oracle_cfg3(cfg3_case):
    """cfg 3: 1920x1080, yolov8l@960 (103 convolutions deep) + HRNet-W48 with the peaked head, BOTH networks in the split family."""
    from eagle_amd.coordinate_model import CoordinateModel
    hs2, yl, frames3, steps = cfg3_case
    cm = CoordinateModel(precision="f32s", detector_precision="f32s", batch=2, frame_hw=(1080, 1920), detector="l", det_imgsz=960, hrnet_state_dict=hs2, detector_state_dict=yl)
    recs = cm.process_records(frames3)
    cm.handle.close()
    tot = []
    for i, (oref, aux) in enumerate(steps):
        tot.append(_f32s_record_parity(recs[i], oref, aux, f"f32s cfg3 frame {i}", (1080, 1920), conf_tol=6e-5, box_tol=1e-2, score_tol=3e-5))
    print("f32s parity (cfg3):", tot)
    assert tot[0]["h_identical"] and sum(t["hm_tie"] + t["near_int_box"] for t in tot) <= 1, tot
    assert sum(t["conf_tie"] for t in tot) <= 0.04 * sum(t["dets"] for t in tot), tot


@pytest.mark.parametrize("det", ["f32s", "mixed"])
def test_f32s_ids_identical_with_a_sparse_detector(state_dicts, det):
    """(det = "mixed", round 6: EAGLE_DET_PREC_MIXED — split trunk, exact last C2f per level + Detect, seamed by split_to_f32 — under the same contract.)
    The id contract (detection index in descending-confidence order, cm.py:598-616) with a realistic number of detections: the class
    biases of the synthetic detector lowered until a frame keeps a few dozen boxes, confidences then lie ~1e-2 apart and every id, class,
    integer box and pitch integer must equal the fp32 oracle's with NO admitted exception."""
    from eagle_amd import synth
    from eagle_amd.coordinate_model import CoordinateModel
    from oracle import pipeline
    hs, ys = state_dicts
    hs2, g = _peaked_state_dict(hs)
    ys2 = dict(ys)
    for l in range(3):
        ys2[f"model.22.cv3.{l}.2.bias"] = (ys[f"model.22.cv3.{l}.2.bias"] - np.float32(1.25)).astype(np.float32)
    frames = np.stack([synth.frame(*g["design"]), synth.frame(0, 9), synth.frame(2, 5)])
    cm = CoordinateModel(precision="f32s", detector_precision=det, batch=3, hrnet_state_dict=hs2, detector_state_dict=ys2)
    recs = cm.process_records(frames)
    cm.handle.close()
    ora = pipeline.OracleModel(hs2, ys2, backend="c")
    tot = []
    for i, f in enumerate(frames):
        oref, aux = ora.step(f, i)
        tot.append(_f32s_record_parity(recs[i], oref, aux, f"f32s sparse detector ({det}) frame {i}", (720, 1280), score_tol=3e-5))
    print("f32s parity (sparse detector):", tot)
    assert all(3 <= t["dets"] <= 120 for t in tot), [t["dets"] for t in tot]
    assert sum(t["hm_tie"] + t["near_int_box"] + t["conf_tie"] for t in tot) == 0, tot


# ---- the DEFAULT handle (round 4): key-points in the split family, detector in the exact fp32 family (EagleConfig::det_precision) ---------------
def _default_handle_parity(rec, oref, aux, tag, frame_hw, score_tol=5e-6):
    """north_star's integer contract WITHOUT an admitted exception: every detection field of the record — float box, confidence, class, NMS
    position (= detection-index id, cm.py:598-627), integer box, foot point — equals the fp32 oracle's (OracleModel(backend="c"), pinned to the
    reference) under np.array_equal; heat-map maxima, key-point pixels, synthesised points and H_valid equal, H bit-identical, pitch integers equal."""
    from eagle_amd import records
    from eagle_amd.pitch import INTERSECTION_TO_PITCH_POINTS
    dets = aux["dets"]
    n = int(rec["n_det"])
    assert n == len(dets), f"{tag}: {n} detections vs {len(dets)}"
    got = np.stack([rec["det"][k][:n] for k in ("x1", "y1", "x2", "y2", "conf")], 1)
    assert np.array_equal(got, dets[:, :5].astype(np.float32)), f"{tag}: float boxes / confidences differ at {np.argwhere(got != dets[:, :5])[:3]}"
    cls = dets[:, 5].astype(np.int32)
    assert np.array_equal(rec["det"]["cls"][:n], cls), f"{tag}: classes differ"
    ids = np.full(n, -1, np.int32)                          # cm.py:598-627: persons keyed by detection index, balls by enumerate index
    ids[(cls == 0) | (cls == 1)] = np.nonzero((cls == 0) | (cls == 1))[0]
    ids[cls == 2] = np.arange(int((cls == 2).sum()))
    assert np.array_equal(rec["det"]["id"][:n], ids), f"{tag}: ids differ"
    fh, fw = frame_hw
    bi = dets[:, :4].astype(np.int64)                       # astype(int) truncation (cm.py:600), persons clipped to the frame
    person = (cls == 0) | (cls == 1)
    bi[person, 0::2] = np.clip(bi[person, 0::2], 0, fw - 1); bi[person, 1::2] = np.clip(bi[person, 1::2], 0, fh - 1)
    gi = np.stack([rec["det"][k][:n] for k in ("bx1", "by1", "bx2", "by2")], 1).astype(np.int64)
    assert np.array_equal(gi, bi), f"{tag}: integer boxes differ at {np.argwhere(gi != bi)[:3]}"
    # key-point half (split family): integers identical, scores to a few 1e-6
    assert np.array_equal(rec["hm_idx"], aux["hm_idx"]), f"{tag}: heat-map maxima differ on channels {np.nonzero(rec['hm_idx'] != aux['hm_idx'])[0]}"
    assert float(np.abs(rec["hm_score"].astype(np.float64) - aux["hm_score"]).max()) <= score_tol
    allkp = {INTERSECTION_TO_PITCH_POINTS[int(k["label"])]: (int(k["x"]), int(k["y"])) for k in rec["kp"][: int(rec["n_kp"])]}
    assert allkp == {k: (int(v[0]), int(v[1])) for k, v in aux["kp_synth"].items()}, f"{tag}: key-points differ"
    assert bool(rec["H_valid"]) == (aux["H"] is not None), f"{tag}: H validity differs"
    if aux["H"] is not None:
        assert np.array_equal(rec["H"].reshape(3, 3), aux["H"]), f"{tag}: H differs"
    assert not rec["pad"][1], f"{tag}: saturation flag set"
    # the reference-schema record: keys (ids), integer boxes, pitch integers, None-ness identical; confidences bit-equal
    got_d, ref_d = _canon_keep(records.to_reference_dict(rec, 0)), _canon_keep(oref)
    assert got_d["Coordinates"] == ref_d["Coordinates"], f"{tag}: Coordinates differ"
    assert got_d["Keypoints"] == ref_d["Keypoints"] and got_d["Boundaries"] == ref_d["Boundaries"], f"{tag}: Keypoints / Boundaries differ"
    return n


def test_default_handle_dense_detector_is_exception_free_cfg2(state_dicts, frames, oracle_steps):
    """VERDICT r3 task 1b: the five random-head cfg-2 frames (240 - 300 boxes each, confidences ~1e-3 apart — the detector on which the split
    family swaps a few near-tie ids) through the DEFAULT configuration: no conf_tie / near_int_box branch exists in this comparison."""
    from eagle_amd import lib
    from eagle_amd.coordinate_model import CoordinateModel
    hs, ys = state_dicts
    cm = CoordinateModel(batch=2, hrnet_state_dict=hs, detector_state_dict=ys)          # all defaults: precision f32s, detector exact fp32
    assert cm.handle.cfg.precision == lib.PREC_F32S and cm.handle.cfg.det_precision == lib.PREC_F32 + 1
    recs = cm.process_records(frames)
    cm.handle.close()
    nd = [_default_handle_parity(recs[i], oref, aux, f"default handle cfg2 frame {i}", (720, 1280)) for i, (oref, aux) in enumerate(oracle_steps)]
    print("default handle, cfg2, detections per frame:", nd)
    assert sum(nd) > 1000, nd


@pytest.fixture(scope="module")
def cfg3_case():
    """Three 1920x1080 frames, yolov8l@960 + HRNet with the peaked head, and the fp32 oracle's steps (shared by the cfg-3 tests)."""
    from eagle_amd import synth, weights
    from oracle import pipeline
    hs, yl = weights.make_hrnet_state_dict(0), weights.make_yolo_state_dict("l", 0)
    hs2, g = _peaked_state_dict(hs)
    frames3 = np.stack([synth.frame(int(g["design"][0]), int(g["design"][1]), 1080, 1920), synth.frame(0, 9, 1080, 1920), synth.frame(2, 5, 1080, 1920)])
    ora = pipeline.OracleModel(hs2, yl, variant="l", imgsz=960, backend="c")
    return hs2, yl, frames3, [ora.step(f, i) for i, f in enumerate(frames3)]


def test_default_handle_dense_detector_is_exception_free_cfg3(cfg3_case):
    """The same on cfg 3 (1920x1080, yolov8l@960: 103 convolutions deep, where the split family's confidences move by 4e-5)."""
    from eagle_amd.coordinate_model import CoordinateModel
    hs2, yl, frames3, steps = cfg3_case
    cm = CoordinateModel(batch=2, frame_hw=(1080, 1920), detector="l", det_imgsz=960, hrnet_state_dict=hs2, detector_state_dict=yl)
    recs = cm.process_records(frames3)
    cm.handle.close()
    nd = [_default_handle_parity(recs[i], oref, aux, f"default handle cfg3 frame {i}", (1080, 1920), score_tol=3e-5) for i, (oref, aux) in enumerate(steps)]
    print("default handle, cfg3, detections per frame:", nd)
    assert sum(nd) > 100, nd


def test_default_handle_on_a_small_frame_size(state_dicts):
    """The same exception-free comparison on 640x360 frames (another letter-box geometry, the key-point resize enlarges instead of shrinking; batch 3 with
    two frames: a ragged batch through the stacked exact tiling)."""
    from eagle_amd import synth
    from eagle_amd.coordinate_model import CoordinateModel
    from oracle import pipeline
    hs, ys = state_dicts
    fr = np.stack([synth.frame(0, 3, 360, 640), synth.frame(1, 8, 360, 640)])
    cm = CoordinateModel(batch=3, frame_hw=(360, 640), hrnet_state_dict=hs, detector_state_dict=ys)
    recs = cm.process_records(fr)
    cm.handle.close()
    ora = pipeline.OracleModel(hs, ys, backend="c")
    nd = []
    for i, f in enumerate(fr):
        oref, aux = ora.step(f, i)
        nd.append(_default_handle_parity(recs[i], oref, aux, f"default handle 360p frame {i}", (360, 640)))
    assert sum(nd) > 0, nd


def test_default_handle_detector_half_equals_the_exact_family(state_dicts, frames):
    """GPU against GPU: every detection field of the default handle is byte-identical to the exact (fp32) handle's."""
    from eagle_amd.coordinate_model import CoordinateModel
    hs, ys = state_dicts
    a = CoordinateModel(batch=3, hrnet_state_dict=hs, detector_state_dict=ys)
    ra = a.process_records(frames); a.handle.close()
    b = CoordinateModel(precision="f32", batch=3, hrnet_state_dict=hs, detector_state_dict=ys)
    rb = b.process_records(frames); b.handle.close()
    assert ra["n_det"].tolist() == rb["n_det"].tolist() and ra["n_candidates"].tolist() == rb["n_candidates"].tolist()
    for f in ("x1", "y1", "x2", "y2", "conf", "cls", "bx1", "by1", "bx2", "by2", "id", "foot_x", "foot_y", "reported"):
        assert ra["det"][f].tobytes() == rb["det"][f].tobytes(), f


# ---- f32s fails loudly (round 4): the split format clips at +-4094 and says so ---------------------------------------------------------------
def test_f32s_saturation_is_reported_not_silent(state_dicts, frames):
    """VERDICT r3 task 3a.  Stem weights scaled x4096 push activations beyond the split format's range: the call must return EAGLE_E_RANGE
    (EagleRangeError), count the clipped stores, flag the frames (pad[1]) and still hand the records over; allow_saturation turns the error
    into flags only; the unscaled network reports 0 / 0 and no flag."""
    from eagle_amd import lib
    from eagle_amd.coordinate_model import CoordinateModel
    hs, ys = state_dicts
    ok = CoordinateModel(batch=2, hrnet_state_dict=hs, detector_state_dict=ys)
    r = ok.process_records(frames[:3])
    t = ok.handle.timings()
    assert (t.sat_events, t.sat_frames) == (0, 0) and not r["pad"][:, 1].any()
    ok.handle.close()
    hot = dict(hs)
    k = "unnormalized_model.0.conv1.weight"
    hot[k] = (hs[k] * np.float32(4096.0)).astype(np.float32)
    cm = CoordinateModel(batch=2, hrnet_state_dict=hot, detector_state_dict=ys)
    with pytest.raises(lib.EagleRangeError) as ei:
        cm.process_records(frames[:3])
    assert "4094" in str(ei.value) and ei.value.records is not None and len(ei.value.records) == 3
    t = cm.handle.timings()
    assert t.sat_events > 0 and t.sat_frames == 3, (t.sat_events, t.sat_frames)
    assert ei.value.records["pad"][:, 1].all()
    # the exact-family detector half of the flagged records is untouched by the key-point network's clipping
    assert ei.value.records["n_det"].tolist() == r["n_det"].tolist()
    cm.handle.close()
    cm2 = CoordinateModel(batch=2, hrnet_state_dict=hot, detector_state_dict=ys, allow_saturation=True)
    r2 = cm2.process_records(frames[:3])
    t2 = cm2.handle.timings()
    assert r2["pad"][:, 1].all() and t2.sat_events == t.sat_events and t2.sat_frames == 3
    cm2.handle.close()
    # a fp32 handle has no such range: same weights, no report
    cm3 = CoordinateModel(precision="f32", batch=2, hrnet_state_dict=hot, detector_state_dict=ys)
    r3 = cm3.process_records(frames[:1])
    assert cm3.handle.timings().sat_events == 0 and not r3["pad"][:, 1].any()
    cm3.handle.close()


def test_f32s_full_size_clip_properties(state_dicts):
    """BASELINE.json's configs[1] at full size in the benchmarked family (1000 frames of 1280x720, device batch 50, the bench's clip: 20 distinct
    frames tiled): every copy of a frame yields byte-identical records wherever it sits in the clip and in its batch, a different device batch
    (37: ragged last batch) yields the same records, the host-fed path equals the resident path, and a hipGraph replay changes nothing."""
    from eagle_amd import synth
    from eagle_amd.coordinate_model import CoordinateModel
    hs, ys = state_dicts
    base = np.stack([synth.frame(0, t) for t in range(20)])
    clip = np.ascontiguousarray(np.tile(base, (50, 1, 1, 1)))
    a = CoordinateModel(precision="f32s", batch=50, hrnet_state_dict=hs, detector_state_dict=ys)
    ra = a.process_records(clip)                                  # host frames: eagle_process_frames
    d = a.handle.upload(clip)
    rd = np.zeros(len(clip), ra.dtype)
    a.handle.process_device(d, len(clip), rd)                     # resident frames: eagle_process_device_frames
    a.handle.free(d); a.handle.close()
    assert ra.tobytes() == rd.tobytes()
    for t in range(20):
        first = ra[t].tobytes()
        assert all(ra[k].tobytes() == first for k in range(t, 1000, 20)), t
    b = CoordinateModel(precision="f32s", batch=37, hrnet_state_dict=hs, detector_state_dict=ys, use_graph=True)
    rb = b.process_records(clip[:148])
    rb2 = b.process_records(clip[:148])                           # graph replay
    b.handle.close()
    assert ra[:148].tobytes() == rb.tobytes() == rb2.tobytes()


def test_f32_family_is_batch_and_position_invariant_at_the_bench_batch(state_dicts):
    """The exact family tiles a batch as ONE stacked image since round 4 (tiles straddle frames): its records must not depend on the device batch or on
    a frame's position in it.  53 frames (4 distinct, tiled) through a batch-50 handle (one full batch + a ragged one of 3) against a batch-3 handle:
    byte-identical records, every copy of a frame identical wherever it sits."""
    from eagle_amd import synth
    from eagle_amd.coordinate_model import CoordinateModel
    hs, ys = state_dicts
    base = np.stack([synth.frame(0, 1), synth.frame(0, 20), synth.noise_frame(3), synth.frame(2, 5)])
    clip = np.ascontiguousarray(np.tile(base, (14, 1, 1, 1))[:53])
    a = CoordinateModel(precision="f32", batch=50, hrnet_state_dict=hs, detector_state_dict=ys)
    ra = a.process_records(clip); a.handle.close()
    b = CoordinateModel(precision="f32", batch=3, hrnet_state_dict=hs, detector_state_dict=ys)
    rb = b.process_records(clip[:9]); b.handle.close()
    assert ra[:9].tobytes() == rb.tobytes()
    for k in range(4):
        first = ra[k].tobytes()
        assert all(ra[j].tobytes() == first for j in range(k, 53, 4)), k


def test_cfg3_full_size_clip_properties():
    """BASELINE.json's configs[2] at full size through the DEFAULT handle (1000 frames of 1920x1080, yolov8l@960 in the exact family + HRNet in the
    split family, device batch 25 — the bench's cfg3 run): every copy of a frame yields byte-identical records wherever it sits in the clip and in its
    batch, a different device batch (13: ragged last batch) yields the same records, the host-fed path equals the resident path, no frame reports a
    clipped activation, and the records are sane (detections in descending confidence, ids by the detection-index rule)."""
    from eagle_amd import synth, weights
    from eagle_amd.coordinate_model import CoordinateModel
    hs, yl = weights.make_hrnet_state_dict(0), weights.make_yolo_state_dict("l", 0)
    base = np.stack([synth.frame(0, t, 1080, 1920) for t in range(10)])
    clip = np.ascontiguousarray(np.tile(base, (100, 1, 1, 1)))
    a = CoordinateModel(batch=25, frame_hw=(1080, 1920), detector="l", det_imgsz=960, hrnet_state_dict=hs, detector_state_dict=yl)
    ra = a.process_records(clip)                                  # host frames: eagle_process_frames
    d = a.handle.upload(clip)
    rd = np.zeros(len(clip), ra.dtype)
    a.handle.process_device(d, len(clip), rd)                     # resident frames: eagle_process_device_frames
    a.handle.free(d)
    t = a.handle.timings()
    a.handle.close()
    assert ra.tobytes() == rd.tobytes()
    assert (t.sat_events, t.sat_frames) == (0, 0) and not ra["pad"][:, 1].any()
    for k in range(10):
        first = ra[k].tobytes()
        assert all(ra[j].tobytes() == first for j in range(k, 1000, 10)), k
    for r in ra[:10]:
        n = int(r["n_det"])
        assert n > 0 and np.all(np.diff(r["det"]["conf"][:n]) <= 0)
        persons = np.isin(r["det"]["cls"][:n], (0, 1))
        assert np.array_equal(r["det"]["id"][:n][persons], np.nonzero(persons)[0])
    b = CoordinateModel(batch=13, frame_hw=(1080, 1920), detector="l", det_imgsz=960, hrnet_state_dict=hs, detector_state_dict=yl)
    rb = b.process_records(clip[:100])
    b.handle.close()
    assert ra[:100].tobytes() == rb.tobytes()


def test_f32s_batch_and_position_invariance(state_dicts, frames):
    from eagle_amd.coordinate_model import CoordinateModel
    hs, ys = state_dicts
    a = CoordinateModel(precision="f32s", batch=1, hrnet_state_dict=hs, detector_state_dict=ys)
    ra = a.process_records(frames)
    a.handle.close()
    b = CoordinateModel(precision="f32s", batch=4, hrnet_state_dict=hs, detector_state_dict=ys)
    rb = b.process_records(frames[::-1])[::-1]
    b.handle.close()
    assert ra.tobytes() == rb.tobytes()


def test_small_batch_mode_records_equal_the_large_batch_handles(state_dicts, frames):
    """Round 5 (VERDICT r4 task 4): handles with batch <= EAGLE_SMALL_BATCH run in small-batch mode by default (the network phase replayed as a hipGraph,
    HRNet's branches and the detector on their own streams).  Same kernels, another launch mechanism: the records of the DEFAULT handle at B = 1, 2, 4, 8
    must be byte-identical to a B = 17 handle's (plain launches, one stream per network) and to a B = 2 handle with both switches off; the resolved
    configuration is visible through eagle_get_config."""
    from eagle_amd import lib
    from eagle_amd.coordinate_model import CoordinateModel
    hs, ys = state_dicts
    big = CoordinateModel(batch=17, hrnet_state_dict=hs, detector_state_dict=ys, use_graph=False, multi_stream=False)        # plain launches, one stream per network
    assert (big.handle.cfg.use_graph, big.handle.cfg.multi_stream) == (0, 0)
    ref = big.process_records(frames)
    big.handle.close()
    dflt = CoordinateModel(batch=17, hrnet_state_dict=hs, detector_state_dict=ys)                           # round 6: branch streams at every batch, graph replay up to 32 frames per step
    assert (dflt.handle.cfg.use_graph, dflt.handle.cfg.multi_stream) == (1, 1)
    assert dflt.process_records(frames).tobytes() == ref.tobytes()
    dflt.handle.close()
    for B in (1, 2, 4, 8, 12):
        m = CoordinateModel(batch=B, hrnet_state_dict=hs, detector_state_dict=ys)
        assert (m.handle.cfg.use_graph, m.handle.cfg.multi_stream) == (1, 1), B       # (EAGLE_SMALL_BATCH = 32 since round 6)
        got = m.process_records(frames)                     # 5 frames: ragged last step at B = 2, 4, 8 (another graph instance)
        again = m.process_records(frames[:1])               # a second call with another frame count on the same handle
        m.handle.close()
        assert got.tobytes() == ref.tobytes(), f"small-batch mode at B = {B} differs from the B = 17 handle"
        assert again.tobytes() == ref[:1].tobytes()
    p = CoordinateModel(batch=2, hrnet_state_dict=hs, detector_state_dict=ys, use_graph=False, multi_stream=False)
    assert (p.handle.cfg.use_graph, p.handle.cfg.multi_stream) == (0, 0)
    assert p.process_records(frames).tobytes() == ref.tobytes()
    p.handle.close()
    # use_graph = 2 on a call of three steps (5 frames through batch 2: 2 + 2 + 1 — replayed, with a re-capture for the ragged last step) and of one step (plain launches)
    q = lib.Handle(batch=2, use_graph=2, multi_stream=0)
    from eagle_amd import weights
    weights.load_into(q, [hs, ys])
    assert q.process(frames).tobytes() == ref.tobytes() and q.process(frames[:2]).tobytes() == ref[:2].tobytes()
    q.close()


# ---- the fp16 (fast) family on trial against the fp32 oracle (NOT its own fp16-emulating oracle) --------------------------------------------------
def _f16_vs_fp32_counters(rec, oref, aux, frame_hw):
    """Counts, per frame, how the fast family's record differs from the fp32 oracle's (the reference's arithmetic): every integer field,
    nothing admitted a priori.  Returns the counters; the caller prints and bounds them."""
    from eagle_amd.pitch import INTERSECTION_TO_PITCH_POINTS
    c = dict(hm_idx=int((rec["hm_idx"] != aux["hm_idx"]).sum()), kp_same=False, h_valid_same=bool(rec["H_valid"]) == (aux["H"] is not None), h_rel=None,
             n_det=(int(rec["n_det"]), len(aux["dets"])), matched=0, cls_same=0, int_box_same=0, id_same=0, max_conf_dev=0.0, pitch_checked=0, pitch_int_same=0, max_pitch_float_dev=0.0)
    kp = {INTERSECTION_TO_PITCH_POINTS[int(k["label"])]: (int(k["x"]), int(k["y"])) for k in rec["kp"][: int(rec["n_kp"])]}
    c["kp_same"] = kp == {k: (int(v[0]), int(v[1])) for k, v in aux["kp_synth"].items()}
    if aux["H"] is not None and rec["H_valid"]:
        c["h_rel"] = float(np.abs(rec["H"].reshape(3, 3) - aux["H"]).max() / np.abs(aux["H"]).max())
    n, dets = int(rec["n_det"]), aux["dets"]
    if n == 0 or len(dets) == 0:
        return c
    g = np.stack([rec["det"][k][:n] for k in ("x1", "y1", "x2", "y2")], 1)
    fh, fw = frame_hw
    for i, d in enumerate(dets):
        io = _iou(d[:4], g)
        j = int(io.argmax())
        if io[j] < 0.9:
            continue
        c["matched"] += 1
        c["cls_same"] += int(int(rec["det"]["cls"][j]) == int(d[5]))
        c["id_same"] += int(j == i)
        c["max_conf_dev"] = max(c["max_conf_dev"], abs(float(rec["det"]["conf"][j]) - float(d[4])))
        bi = np.array([int(d[0]), int(d[1]), int(d[2]), int(d[3])])
        if int(d[5]) in (0, 1):
            bi = np.array([min(max(bi[0], 0), fw - 1), min(max(bi[1], 0), fh - 1), min(max(bi[2], 0), fw - 1), min(max(bi[3], 0), fh - 1)])
        c["int_box_same"] += int(all(int(rec["det"][k][j]) == b for k, b in zip(("bx1", "by1", "bx2", "by2"), bi)))
    # pitch coordinates of the reported persons: floats at north_star's 1e-3 (relative to the pitch length) even when H differs in its last bits
    for cname in ("Player", "Goalkeeper"):
        for oid, o in oref["Coordinates"].get(cname, {}).items():
            if o.get("_pitch_float") is None or int(oid) >= len(dets):
                continue
            io = _iou(dets[int(oid)][:4], g)
            j = int(io.argmax())
            gd = rec["det"][j]
            ofoot = [int((o["BBox"][0] + o["BBox"][2]) / 2), o["BBox"][3]]
            if io[j] < 0.9 or [int(gd["foot_x"]), int(gd["foot_y"])] != ofoot or not gd["in_bounds"]:
                continue
            c["pitch_checked"] += 1
            c["pitch_int_same"] += int([int(gd["pitch_x"]), int(gd["pitch_y"])] == [int(v) for v in o["Transformed_Coordinates"]])
            c["max_pitch_float_dev"] = max(c["max_pitch_float_dev"], abs(float(gd["pitch_xf"]) - o["_pitch_float"][0]), abs(float(gd["pitch_yf"]) - o["_pitch_float"][1]))
    return c


def test_f16_family_against_the_fp32_oracle_counters(state_dicts):
    """VERDICT r2 task 2a: the fast family against the fp32 oracle (backend "c", the one pinned to the reference) on the six peaked-head frames.
    Geometry: key-points identical and H solved on >= 5 of 6 frames, H within 1e-3 relative, pitch floats within 1e-3 * 105 m.  Detector: the
    integer-field agreement is COUNTED, printed and bounded — it is what fp16 tensors cost on a random-weight detector with 250 - 300 boxes per
    frame, and why the fp16 family is not the benchmarked one (the f32s tests above are exact on the same frames)."""
    from eagle_amd import synth
    from eagle_amd.coordinate_model import CoordinateModel
    from oracle import pipeline
    hs, ys = state_dicts
    hs2, g = _peaked_state_dict(hs)
    frames = np.stack([synth.frame(*g["design"]), synth.frame(0, 4), synth.frame(0, 9), synth.frame(0, 1), synth.frame(0, 6), synth.frame(0, 14)])
    cm = CoordinateModel(precision="f16", batch=3, hrnet_state_dict=hs2, detector_state_dict=ys)
    recs = cm.process_records(frames)
    cm.handle.close()
    # the mixed handle: key-points in fp16, detector in the split (fp32-grade) family
    cmx = CoordinateModel(precision="f16", detector_precision="f32s", batch=3, hrnet_state_dict=hs2, detector_state_dict=ys)
    recx = cmx.process_records(frames)
    cmx.handle.close()
    ora = pipeline.OracleModel(hs2, ys, backend="c")
    tot, totx = [], []
    for i, f in enumerate(frames):
        oref, aux = ora.step(f, i)
        tot.append(_f16_vs_fp32_counters(recs[i], oref, aux, (720, 1280)))
        totx.append(_f16_vs_fp32_counters(recx[i], oref, aux, (720, 1280)))
        # the mixed handle's detector half equals the f32s family's: detection count, classes, ids as in test_f32s_* (near-ties aside)
        assert totx[-1]["n_det"][0] == totx[-1]["n_det"][1], (i, totx[-1])
        assert totx[-1]["max_conf_dev"] < 5e-6 and totx[-1]["cls_same"] == totx[-1]["matched"] and totx[-1]["int_box_same"] == totx[-1]["matched"], (i, totx[-1])
        assert totx[-1]["id_same"] >= totx[-1]["matched"] - 4, (i, totx[-1])
    print("fp16 family vs fp32 oracle:", tot)
    print("fp16 key-points + f32s detector vs fp32 oracle:", totx)
    for fam in (tot, totx):
        assert sum(t["kp_same"] and t["h_valid_same"] and t["h_rel"] is not None for t in fam) >= 5, fam      # geometry checked on >= 5 of 6 frames
        assert all(t["h_rel"] is None or t["h_rel"] <= 1e-3 for t in fam), fam
        assert all(t["max_pitch_float_dev"] <= 1e-3 * 105 for t in fam) and sum(t["pitch_checked"] for t in fam) > 20, fam
    m = sum(t["matched"] for t in tot)
    assert m >= 0.9 * sum(t["n_det"][1] for t in tot), tot
    assert sum(t["cls_same"] for t in tot) >= 0.98 * m and max(t["max_conf_dev"] for t in tot) < 2.5e-3, tot
    assert sum(t["int_box_same"] for t in tot) >= 0.70 * m, tot          # measured ~ 80 %: one network-input pixel is two frame pixels


def test_mixed_handle_detector_records_equal_the_f32s_family(state_dicts, frames):
    """EagleConfig::det_precision: the detector half of a fp16 handle run in the split family is bit-identical to the f32s handle's detector
    half (same kernels, same tensors), for 1.4 % of the FLOPs."""
    from eagle_amd.coordinate_model import CoordinateModel
    hs, ys = state_dicts
    a = CoordinateModel(precision="f32s", detector_precision="f32s", batch=2, hrnet_state_dict=hs, detector_state_dict=ys)
    ra = a.process_records(frames)
    a.handle.close()
    b = CoordinateModel(precision="f16", detector_precision="f32s", batch=2, hrnet_state_dict=hs, detector_state_dict=ys)
    rb = b.process_records(frames)
    b.handle.close()
    assert ra["n_det"].tolist() == rb["n_det"].tolist()
    for f in ("x1", "y1", "x2", "y2", "conf", "cls", "bx1", "by1", "bx2", "by2", "id"):
        assert ra["det"][f].tobytes() == rb["det"][f].tobytes(), f

"""-m gpu: K17, boxmot's default camera-motion estimator (ECC on the 0.15-scale gray frame; the BotSort of cm.py:66-72 runs it on the frame of
cm.py:577) through the C ABI (eagle_clip_motion_ecc) against oracle/ecc.py: warps, failure handling (identity + the OLD template kept), the
template carried from clip to clip, and the tracker end to end."""
import numpy as np
import pytest

from eagle_amd import lib, synth
from oracle import ecc

pytestmark = pytest.mark.gpu
IDENT = [1, 0, 0, 0, 1, 0]
# float64 sums in a different order than numpy's -> float32 projections may differ in the last bit -> the converged warp agrees to ~1e-5 of a
# small-image pixel; translations are divided by 0.15
ATOL_R, ATOL_T = 2e-6, 2e-3


def _handle():
    from eagle_amd.coordinate_model import CoordinateModel
    return CoordinateModel(batch=1).handle                 # (the clip session needs a finalised handle)


def _gpu_motion(h, frames, carry=False, first=0, count=None):
    d = h.upload(np.stack(frames))
    try:
        h.clip_open(d, len(frames))
        out = h.clip_motion_ecc(first, len(frames) - first if count is None else count, carry=carry, return_ok=True)
        h.clip_close()
    finally:
        h.free(d)
    return out


def _close(a, b):
    a = np.asarray(a).reshape(-1, 6); b = np.asarray(b).reshape(-1, 6)
    return (np.abs(a[:, [0, 1, 3, 4]] - b[:, [0, 1, 3, 4]]).max(initial=0) <= ATOL_R and np.abs(a[:, [2, 5]] - b[:, [2, 5]]).max(initial=0) <= ATOL_T)


def test_small_images_bit_equal_oracle_preprocess():
    """the 0.15-scale gray image (cv2.cvtColor + cv2.resize restated) is integer work: the warp of an identical pair is exactly the identity and
    any difference in the small images would show up as a different warp below; checked directly through a one-iteration alignment of a frame
    with itself (rho = 1 needs bit-equal inputs on both sides only) and through the full comparison of the next test."""
    f = synth.frame(0, 5)
    h = _handle()
    w, ok = _gpu_motion(h, [f, f])
    h.close()
    assert ok.tolist() == [1, 1] and np.abs(w[1] - IDENT).max() < 1e-6


def test_clip_motion_ecc_equals_oracle():
    frames = [synth.frame(0, t) for t in (4, 6, 8, 8, 9)] + [synth.frame(1, 40)]
    black = np.zeros_like(frames[0])
    seq = frames[:3] + [black] + frames[3:] + [255 - frames[-1], synth.frame(1, 41)]
    exp = ecc.clip_motion(seq)
    h = _handle()
    w, ok = _gpu_motion(h, seq)
    w2, ok2 = _gpu_motion(h, seq, first=2, count=2)
    h.close()
    assert np.array_equal(w[0], IDENT)
    # frame 3 is black (NaN correlation) and frame 7 the negative of frame 6 (lambda_d <= 0): cv2 raises, boxmot returns the identity and keeps
    # the old template, so frames 4 and 8 are aligned to frames 2 and 6
    assert ok.tolist() == [1, 1, 1, 0, 1, 1, 1, 0, 1]
    assert np.array_equal(w[3], IDENT) and np.array_equal(w[7], IDENT)
    assert _close(w, exp), np.abs(w - exp).max(axis=0)
    assert np.abs(exp[6] - IDENT).max() > 1.0               # the scene cut is a real alignment problem (tens of iterations), not an identity
    assert _close(w2[0], exp[2]) and ok2.tolist() == [1, 0]


def test_template_is_carried_across_clips_like_boxmots_estimator():
    a = [synth.frame(0, t) for t in (1, 2, 3)]
    b = [synth.frame(0, t) for t in (4, 5)]
    e = ecc.ECC()
    exp = np.stack([e.apply(f).astype(np.float64).reshape(6) for f in a + b])
    h = _handle()
    h.track_open()                                          # a new tracker = a new estimator
    wa, _ = _gpu_motion(h, a, carry=True)
    wb, _ = _gpu_motion(h, b, carry=True)
    wc, _ = _gpu_motion(h, b, carry=False)
    h.track_open()
    wd, _ = _gpu_motion(h, b, carry=True)
    h.close()
    assert _close(np.concatenate([wa, wb]), exp)
    assert np.abs(wb[0] - IDENT).max() > 1e-4              # frame 0 of the second clip was aligned to the last frame of the first
    assert np.array_equal(wc[0], IDENT) and np.array_equal(wd[0], IDENT) and _close(wc[1], exp[4]) and _close(wd[1], exp[4])


def test_full_hd_and_tiny_frames():
    rng = np.random.default_rng(0)
    from scipy.ndimage import gaussian_filter
    tex = gaussian_filter(rng.random((1080 + 64, 1920 + 64)), 6.0)
    tex = ((tex - tex.min()) / (tex.max() - tex.min()) * 255).astype(np.uint8)
    f = [np.repeat(tex[32 + dy:32 + dy + 1080, 32 + dx:32 + dx + 1920, None], 3, 2) for dx, dy in ((0, 0), (13, -6))]
    exp = ecc.clip_motion(f)
    from eagle_amd.coordinate_model import CoordinateModel
    cm = CoordinateModel(batch=1, frame_hw=(1080, 1920))
    w, ok = _gpu_motion(cm.handle, f)
    cm.handle.close()
    assert ok.tolist() == [1, 1] and _close(w, exp) and abs(w[1][2] + 13) < 0.5 and abs(w[1][5] - 6) < 0.5
    cm = CoordinateModel(batch=1, frame_hw=(64, 64))        # 0.15 x 64 = 10 pixels: still an image; below 4 the library refuses
    g = [np.ascontiguousarray(x[:64, :64]) for x in f]
    w, ok = _gpu_motion(cm.handle, g)
    cm.handle.close()
    assert _close(w, ecc.clip_motion(g))


def test_coordinate_model_with_ecc_motion_equals_the_tracker_fed_with_oracle_warps():
    """CoordinateModel(tracker=True, camera_motion="ecc") end to end, both cadences: the records equal those of the same tracker fed with
    oracle/ecc.py's warps (ids, boxes, pitch coordinates)."""
    from eagle_amd import records
    from eagle_amd.coordinate_model import CoordinateModel
    frames = np.stack([synth.frame(0, t) for t in range(6)])
    cm = CoordinateModel(batch=2, tracker=True, camera_motion="ecc", detector_conf=0.2)
    got = cm.get_coordinates(frames, fps=1)
    got_flow = cm.get_coordinates(frames, fps=24, num_keypoint_detection=3)      # the clip-session cadence; the template is carried from the first clip
    cm.handle.close()
    ref = CoordinateModel(batch=2, tracker=True, camera_motion=False, detector_conf=0.2)
    recs = ref.process_records(frames)
    e = ecc.ECC()
    w1 = np.stack([e.apply(f).astype(np.float64).reshape(6) for f in frames])
    ref._track(recs, w1)
    exp = {i: records.to_reference_dict(r, i, 1) for i, r in enumerate(recs)}
    ref.handle.close()
    assert sorted(got) == sorted(exp) == list(range(6))
    n_obj = 0
    for i in exp:                                           # (GPU warps agree with the oracle's to ~1e-4 px: integers may flip by one at a truncation edge)
        for cname in ("Player", "Goalkeeper", "Ball"):
            g, e_ = got[i]["Coordinates"].get(cname, {}), exp[i]["Coordinates"].get(cname, {})
            assert set(g) == set(e_), (i, cname, sorted(g), sorted(e_))
            for oid in e_:
                n_obj += 1
                assert np.abs(np.subtract(g[oid]["BBox"], e_[oid]["BBox"])).max() <= 1
                if e_[oid].get("Transformed_Coordinates") is not None:
                    assert np.abs(np.subtract(g[oid]["Transformed_Coordinates"], e_[oid]["Transformed_Coordinates"])).max() <= 1
    assert n_obj > 0
    assert sorted(got_flow) == list(range(6))
    assert np.abs(w1[1:] - IDENT).max() > 1e-3             # the estimator did something on these frames


def test_coordinate_model_carries_the_ecc_template_across_clips_and_resets_it():
    """boxmot's ECC object lives inside the tracker, so clip k+1's first frame is aligned to clip k's last frame.  CoordinateModel used to
    open the tracker lazily AFTER the clip's motion estimate — eagle_track_open forgets the template, so the first clip's template was thrown away
    and the second clip's frame 0 got the identity.  Two clips through get_coordinates (both cadences), the warps the model obtained captured at the
    handle: they equal ONE oracle estimator fed with all frames in order; after reset_tracker() the next clip starts from the identity again."""
    from eagle_amd.coordinate_model import CoordinateModel
    a = [synth.frame(0, t) for t in range(4)]
    b = [synth.frame(0, t) for t in range(9, 13)]             # the camera has moved on between the clips
    for cadence in ({"fps": 1}, {"fps": 24, "num_keypoint_detection": 3}):
        cm = CoordinateModel(batch=2, tracker=True, camera_motion="ecc", detector_conf=0.2)
        seen = []
        inner = cm.handle.clip_motion_ecc

        def spy(*args, **kw):
            w = inner(*args, **kw)
            seen.append(np.array(w[0] if isinstance(w, tuple) else w))
            return w
        cm.handle.clip_motion_ecc = spy
        cm.get_coordinates(np.stack(a), **cadence)
        cm.get_coordinates(np.stack(b), **cadence)
        cm.reset_tracker()
        cm.get_coordinates(np.stack(b), **cadence)
        cm.handle.close()
        assert len(seen) == 3 and all(w.shape == (4, 6) for w in seen), [w.shape for w in seen]
        e = ecc.ECC()
        exp = np.stack([e.apply(f).astype(np.float64).reshape(6) for f in a + b])
        assert _close(seen[0], exp[:4]) and _close(seen[1], exp[4:]), (cadence, seen[1][0], exp[4])
        assert np.abs(exp[4] - IDENT).max() > 1e-3, "the clips must differ at the boundary for this test to mean anything"
        assert np.abs(seen[1][0] - IDENT).max() > 1e-3       # clip 2, frame 0: aligned to clip 1's last frame, not the identity
        e2 = ecc.ECC()
        assert _close(seen[2], np.stack([e2.apply(f).astype(np.float64).reshape(6) for f in b]))      # fresh estimator after the reset
        assert np.array_equal(seen[2][0], IDENT)


def test_bad_arguments_and_degenerate_ranges():
    from eagle_amd.coordinate_model import CoordinateModel
    h = _handle()
    with pytest.raises(lib.EagleError):
        h.clip_motion_ecc(0, 1)                             # no open clip session
    f = [synth.frame(0, t) for t in (1, 2, 3)]
    d = h.upload(np.stack(f))
    try:
        h.clip_open(d, 3)
        w0, ok0 = h.clip_motion_ecc(0, 0, return_ok=True)  # empty range
        assert w0.shape == (0, 6) and ok0.shape == (0,)
        with pytest.raises(lib.EagleError):
            h.clip_motion_ecc(2, 2)                         # beyond the clip
        w_all = h.clip_motion_ecc(0, 3)
        w_last = h.clip_motion_ecc(2, 1)                    # a range that starts inside the clip is aligned to its predecessor
        assert np.array_equal(w_last[0], w_all[2])
        h.clip_close()
        h.clip_open(d, 1)                                   # a one-frame clip: identity, and the frame becomes the carried template
        h.track_open()
        assert np.array_equal(h.clip_motion_ecc(0, 1, carry=True)[0], IDENT)
        h.clip_close()
        h.clip_open(d, 3)
        w_c = h.clip_motion_ecc(0, 3, carry=True)
        h.clip_close()
        assert np.abs(w_c[0] - IDENT).max() < 1e-6           # frame 0 against the carried copy of itself: the identity (rho = 1 at once)
        assert np.array_equal(w_c[1:], w_all[1:])
    finally:
        h.free(d); h.close()
    with pytest.raises(ValueError):
        CoordinateModel(batch=1, tracker=True, camera_motion="dense")


def test_reference_tracker_configuration_is_repeatable_over_a_long_clip():
    """The reference's whole BotSort configuration — association + OSNet appearance + ECC camera motion — over a 120-frame clip in the 25-fps
    cadence (HRNet every 8th frame, optical flow in between): two independent models give identical dictionaries (no run-to-run noise in any of
    the three stages), ids are positive integers that persist (far fewer ids than detections), every frame is present."""
    from eagle_amd.coordinate_model import CoordinateModel
    frames = np.stack([synth.frame(0, t) for t in range(120)])
    outs = []
    for _ in range(2):
        cm = CoordinateModel(batch=8, tracker=True, reid=True, camera_motion="ecc", detector_conf=0.2)
        outs.append(cm.get_coordinates(frames, fps=25, num_homography=1, num_keypoint_detection=3))
        cm.handle.close()
    assert sorted(outs[0]) == list(range(120))
    assert outs[0] == outs[1]
    ids, n_obj = set(), 0
    for i in outs[0]:
        for cname in ("Player", "Goalkeeper"):
            for oid in outs[0][i]["Coordinates"].get(cname, {}):
                assert isinstance(oid, int) and oid >= 0
                ids.add(oid); n_obj += 1
    assert n_obj > 0 and len(ids) * 4 <= n_obj          # tracks persist: on average an id is seen on at least four frames

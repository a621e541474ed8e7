"""CPU tests: the oracle's HRNet restatement against golden vectors produced by the REFERENCE's own module
(tests/golden/make_golden.py imports eagle/models/keypoint_hrnet.py in the build container)."""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.fixture(scope="module")
def golden():
    return np.load(os.path.join(HERE, "golden", "hrnet_golden.npz"))


@pytest.fixture(scope="module")
def small_input():
    return np.random.default_rng(11).standard_normal((2, 100, 148, 3)).astype(np.float32)


@pytest.mark.parametrize("backend", ["c", "torch"])
def test_hrnet_logits_match_reference_module(state_dicts, golden, small_input, backend):
    from oracle import nets
    lg = nets.hrnet_logits(state_dicts[0], small_input, backend=backend)
    ref = golden["small_logits"]
    assert lg.shape == ref.shape == (2, 25, 37, 57)
    assert np.abs(lg - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-5      # fp32 rounding only (different summation order)


def test_get_keypoints_match_reference(state_dicts, golden, small_input):
    """a3: first-maximum index, normalised coordinates and scores of KeypointModel.get_keypoints (kh.py:575-595)."""
    from oracle import host, nets, prims
    lg = nets.hrnet_logits(state_dicts[0], small_input, backend="c")
    for f in range(2):
        idx, score = prims.heatmap_argmax(lg[f], 57)
        dec = host.decode_heatmaps(idx, score, 25, 37)
        ref = [tuple(r) for r in golden["small_kp"][f] if r[0] >= 0]
        assert [d[0] for d in dec] == [int(r[0]) for r in ref]
        for d, r in zip(dec, ref):
            assert d[1] == r[1] and d[2] == r[2]                             # same pixel -> identical float64 x_n, y_n
            assert abs(d[3] - r[3]) < 5e-6


def test_f16_emulation_stays_close(state_dicts, small_input):
    from oracle import nets
    a = nets.hrnet_logits(state_dicts[0], small_input[:1], backend="c")
    b = nets.hrnet_logits(state_dicts[0], small_input[:1], backend="c", f16=True)
    assert np.abs(a - b).max() < 0.02 * np.abs(a).max()


def test_full_frame_against_reference(state_dicts, golden):
    """Full 540x960 frame: strided logits and get_keypoints of the reference module vs the exact-order C oracle."""
    from eagle_amd import synth
    from oracle import host, nets, prims
    x = host.preprocess_keypoints(synth.frame(0, 0))
    lg = nets.hrnet_logits(state_dicts[0], x, backend="c")
    ref = golden["full_logits_strided"]
    assert np.abs(lg[:, ::8, ::8] - ref).max() <= 2e-5 * np.abs(ref).max() + 1e-5
    idx, score = prims.heatmap_argmax(lg[0], 57)
    dec = host.decode_heatmaps(idx, score, 135, 240)
    refk = [tuple(r) for r in golden["full_kp"] if r[0] >= 0]
    assert [d[0] for d in dec] == [int(r[0]) for r in refk]
    same = sum(1 for d, r in zip(dec, refk) if d[1] == r[1] and d[2] == r[2])
    assert same >= len(refk) - 1          # fp32 summation-order noise may move at most a near-tie
    for d, r in zip(dec, refk):
        assert abs(d[3] - r[3]) < 1e-5


def test_yolo_backends_agree(state_dicts):
    """ultralytics is absent (parity unpinned): the exact-order C backend and the torch backend must agree."""
    from oracle import nets
    x = np.random.default_rng(3).uniform(0, 1, (1, 96, 160, 3)).astype(np.float32)
    a = nets.yolo_heads(state_dicts[1], x, "n", backend="c")
    b = nets.yolo_heads(state_dicts[1], x, "n", backend="torch")
    for (ba, ca), (bb, cb) in zip(a, b):
        assert ba.shape == bb.shape and ba.shape[-1] == 64 and ca.shape[-1] == 5
        assert np.abs(ba - bb).max() < 2e-4 and np.abs(ca - cb).max() < 2e-4
    rows = nets.yolo_decode(a)
    assert rows.shape == (12 * 20 + 6 * 10 + 3 * 5, 9) and np.isfinite(rows).all()

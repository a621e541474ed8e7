"""CPU: oracle/ecc.py (boxmot's default camera-motion estimator: cv2.findTransformECC on the 0.15-scale gray frame, restated from the
publication; cv2 / boxmot absent -> parity unpinned) — the primitives against hand-computed values and the estimator against known motions."""
import numpy as np
import pytest
from scipy.ndimage import affine_transform, gaussian_filter, zoom

from oracle import ecc

F32 = np.float32
IDENT = np.array([[1, 0, 0], [0, 1, 0]], F32)


def _texture(seed=0, h=720, w=1280, margin=100):
    rng = np.random.default_rng(seed)
    base = gaussian_filter(rng.random(((h + 2 * margin) // 8 + 2, (w + 2 * margin) // 8 + 2)), 1.0)
    big = zoom(base, 8, order=1)
    return (big - big.min()) / (big.max() - big.min()) * 255.0


def _view(big, dx, dy, h=720, w=1280, margin=100):
    g = big[margin + dy:margin + dy + h, margin + dx:margin + dx + w].astype(np.uint8)
    return np.repeat(g[:, :, None], 3, axis=2)


def test_gradients_hand_computed():
    img = np.array([[10, 20, 40], [0, 100, 50], [7, 7, 7]], np.uint8)
    gx, gy = ecc.gradients(img)
    assert gx.tolist() == [[0, 15, 0], [0, 25, 0], [0, 0, 0]]            # reflect-101: both neighbours of a border pixel are the same pixel
    assert gy.tolist() == [[0, 0, 0], [-1.5, -6.5, -16.5], [0, 0, 0]]


def test_warp_identity_integer_shift_and_border():
    rng = np.random.default_rng(1)
    src = rng.integers(0, 255, (9, 13)).astype(F32)
    assert np.array_equal(ecc.warp_linear(src, IDENT, 9, 13), src)
    M = np.array([[1, 0, 2], [0, 1, -1]], F32)                            # WARP_INVERSE_MAP: dst(x, y) = src(x + 2, y - 1)
    out = ecc.warp_linear(src, M, 9, 13)
    assert np.array_equal(out[1:, :11], src[:-1, 2:]) and (out[0] == 0).all() and (out[:, 11:] == 0).all()
    mask = ecc.warp_mask(M, 9, 13, 9, 13)
    assert mask[1:, :11].all() and not mask[0].any() and not mask[:, 11:].any()
    # half-pixel shift: the 1/32-pixel table gives exactly the mean of the two neighbours; the last column blends with the zero border
    M = np.array([[1, 0, 0.5], [0, 1, 0]], F32)
    out = ecc.warp_linear(src, M, 9, 13)
    assert np.array_equal(out[:, :-1], F32(0.5) * src[:, :-1] + F32(0.5) * src[:, 1:]) and np.array_equal(out[:, -1], F32(0.5) * src[:, -1])
    # coordinates are quantised to 1/32 pixel (round_delta = 16 of 1024): 0.49 -> 16/32
    assert np.array_equal(ecc.warp_linear(src, np.array([[1, 0, 0.49], [0, 1, 0]], F32), 9, 13), out)


def test_inv3_against_numpy_and_singular():
    rng = np.random.default_rng(2)
    A = rng.normal(size=(3, 3)).astype(F32); A = A @ A.T + np.eye(3, dtype=F32)
    assert np.allclose(ecc._inv3(A), np.linalg.inv(A.astype(np.float64)), rtol=1e-6, atol=1e-7)
    assert np.array_equal(ecc._inv3(np.ones((3, 3), F32)), np.zeros((3, 3), F32))      # cv::invert: all zeros, no exception


def test_preprocess_shape_and_flat_frames():
    f = np.full((720, 1280, 3), 77, np.uint8)
    s = ecc.preprocess(f)
    assert s.shape == (108, 192) and (s == 77).all()
    assert ecc.preprocess(np.zeros((1080, 1920, 3), np.uint8)).shape == (162, 288)


def test_identical_frames_give_identity_in_two_iterations():
    a = ecc.preprocess(_view(_texture(), 0, 0))
    rho, M, it = ecc.find_transform_ecc(a, a)
    assert it == 2 and abs(rho - 1.0) < 1e-9 and np.abs(M - IDENT).max() < 1e-6


@pytest.mark.parametrize("dx,dy", [(20, -7), (-33, 12), (6, 40)])
def test_recovers_a_pan(dx, dy):
    big = _texture()
    w = ecc.clip_motion([_view(big, 0, 0), _view(big, dx, dy)])
    assert np.array_equal(w[0], [1, 0, 0, 0, 1, 0])
    # the view moves by (dx, dy): content moves by (-dx, -dy) from the previous frame to the current one
    assert abs(w[1][2] + dx) < 0.35 and abs(w[1][5] + dy) < 0.35, w[1]
    assert abs(w[1][1]) < 2e-4 and abs(w[1][0] - 1) < 1e-6


def test_recovers_a_rotation_about_the_image_centre():
    big = _texture(3)
    f0 = _view(big, 0, 0)[:, :, 0].astype(np.float64)
    th = np.deg2rad(0.8)
    c, s = np.cos(th), np.sin(th)
    R = np.array([[c, -s], [s, c]])                        # (row, col) rotation for affine_transform: output(o) = input(R (o - ctr) + ctr)
    ctr = np.array([359.5, 639.5])
    f1 = affine_transform(f0, R, offset=ctr - R @ ctr, order=1, mode="nearest")
    w = ecc.clip_motion([np.repeat(f0.astype(np.uint8)[:, :, None], 3, 2), np.repeat(np.clip(f1, 0, 255).astype(np.uint8)[:, :, None], 3, 2)])[1].reshape(2, 3)
    ang = np.arcsin(w[1, 0])
    assert abs(abs(ang) - th) < 0.06 * th, (ang, th)
    assert np.abs(w[:, :2] @ ctr[::-1] + w[:, 2] - ctr[::-1]).max() < 0.6      # the centre stays where it is


def test_failed_alignment_returns_identity_and_keeps_the_old_template():
    big = _texture(4)
    a, b = _view(big, 0, 0), _view(big, 14, 3)
    black = np.zeros_like(a)
    e = ecc.ECC()
    assert np.array_equal(e.apply(a), IDENT)
    assert np.array_equal(e.apply(black), IDENT)           # zero variance: NaN correlation -> cv2 raises -> identity
    w = e.apply(b)                                         # aligned to frame 0, NOT to the black frame
    assert abs(w[0, 2] + 14) < 0.35 and abs(w[1, 2] + 3) < 0.35
    e2 = ecc.ECC(); e2.apply(a)
    assert np.array_equal(e2.apply(b), w)
    # an inverted frame is anti-correlated: lambda_d <= 0 -> raises -> identity
    e3 = ecc.ECC(); e3.apply(a)
    assert np.array_equal(e3.apply(255 - a), IDENT) and np.array_equal(e3.prev, ecc.preprocess(a))
    assert ecc.find_transform_ecc(ecc.preprocess(a), ecc.preprocess(255 - a)) is None


def test_warp_drives_the_tracker_compensation():
    """The warps have the direction BoT-SORT's multi_gmc expects (previous frame -> current frame): a track state moved by the estimated warp
    lands on the panned detection."""
    from oracle.tracker import BotSortLite
    big = _texture(5)
    dx, dy = 30, -10
    frames = [_view(big, 0, 0), _view(big, dx, dy)]
    w = ecc.clip_motion(frames)
    d0 = np.array([[600, 300, 640, 390, 0.9, 0]], np.float64)
    d1 = d0.copy(); d1[0, [0, 2]] -= dx; d1[0, [1, 3]] -= dy
    tr = BotSortLite()
    tr.update(d0, w[0].reshape(2, 3))
    out = tr.update(d1, w[1].reshape(2, 3))
    assert len(out) == 1 and out[0][4] == 1
    assert np.abs(np.asarray(out[0][:4]) - d1[0, :4]).max() < 1.0

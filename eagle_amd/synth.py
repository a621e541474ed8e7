"""Deterministic synthetic broadcast frames (no video decode and no dataset exist here).

``frame(seed, t)`` -> uint8 HWC **BGR** (the layout ``cv2.VideoCapture`` hands the reference,
eagle/utils/io.py:16-27): striped pitch-green background, the pitch markings projected through a smooth
synthetic camera homography, ~25 coloured rectangles (players), one small disc (ball) and low-amplitude
PCG64 noise.  ``noise_frame`` is the pure-noise variant used for kernel known-answer tests."""
import numpy as np

from .pitch import LANDMARKS

_SEGMENTS = (  # pitch markings as pairs of landmark indices (straight lines only; arcs sampled separately)
    (12, 13), (28, 29), (12, 28), (13, 29), (14, 15),
    (10, 8), (8, 9), (9, 11), (6, 4), (4, 5), (5, 7),
    (18, 16), (16, 17), (17, 19), (22, 20), (20, 21), (21, 23),
)


def camera(seed, t):
    """World (metres) -> image homography for frame t: a slowly panning/zooming broadcast view."""
    r = np.random.Generator(np.random.PCG64([seed, 77]))
    ph = r.uniform(0, 2 * np.pi, 3)
    cx = 52.5 + 18.0 * np.sin(0.011 * t + ph[0])
    zoom = 13.0 + 3.0 * np.sin(0.007 * t + ph[1])
    tilt = 0.0045 + 0.001 * np.sin(0.005 * t + ph[2])
    # image = K * [world - centre], with a perspective term in y (far touch-line is narrower)
    A = np.array([[zoom, 0.35 * zoom, 640.0 - zoom * cx - 0.35 * zoom * 34.0],
                  [0.0, -0.62 * zoom, 400.0 + 0.62 * zoom * 34.0],
                  [0.0, tilt, 1.0 - tilt * 34.0]], np.float64)
    return A


def project(Hm, pts):
    p = np.concatenate([pts, np.ones((len(pts), 1))], 1) @ Hm.T
    return p[:, :2] / p[:, 2:3]


def visible_landmarks(seed, t, h=720, w=1280):
    """Integer pixel positions of the on-plane landmarks inside the frame: {index: (x, y)}."""
    Hm = camera(seed, t) @ np.diag([1.0, 1.0, 1.0])
    out = {}
    for i, _, x, y, z in LANDMARKS:
        if z != 0.0:
            continue
        u, v = project(Hm, np.array([[x, y]]))[0]
        u *= w / 1280.0
        v *= h / 720.0
        if 0 <= u < w and 0 <= v < h:
            out[i] = (int(u), int(v))
    return out


def _stamp(img, xs, ys, color, rad=1):
    h, w = img.shape[:2]
    for dy in range(-rad, rad + 1):
        for dx in range(-rad, rad + 1):
            x = xs + dx
            y = ys + dy
            m = (x >= 0) & (x < w) & (y >= 0) & (y < h)
            img[y[m], x[m]] = color


_PLAYER_COLOURS = np.where(np.arange(25)[:, None] < 12, np.array([[40, 40, 220]]), np.array([[220, 200, 40]]))
_PLAYER_COLOURS[24] = (20, 20, 20)


def player_boxes(seed, t, h=720, w=1280, _with_height=False):
    """The rectangles frame() draws for the 25 people: [(k, x0, y0, x1, y1)], clipped to the frame (k < 12: first team, 12..23: second
    team, 24: referee).  Ground truth for the track / team-colour tests."""
    Hm = camera(seed, t)
    sx, sy = w / 1280.0, h / 720.0
    rp = np.random.Generator(np.random.PCG64([seed, 5]))
    base = np.stack([rp.uniform(5, 100, 25), rp.uniform(4, 64, 25)], 1)
    vel = rp.normal(0, 0.03, (25, 2))
    feet = project(Hm, base + vel * t)
    out = []
    for k in range(25):
        fx, fy = feet[k, 0] * sx, feet[k, 1] * sy
        ph_ = int(max(18, 0.09 * fy + 8) * sy)
        pw_ = max(6, ph_ // 3)
        x0, x1 = int(fx - pw_ // 2), int(fx + pw_ // 2)
        y0, y1 = int(fy - ph_), int(fy)
        x0, x1, y0, y1 = max(x0, 0), min(x1, w), max(y0, 0), min(y1, h)
        if x1 > x0 and y1 > y0:
            out.append((k, x0, y0, x1, y1, ph_) if _with_height else (k, x0, y0, x1, y1))
    return out


def frame(seed, t, h=720, w=1280):
    r = np.random.Generator(np.random.PCG64([seed, 1000 + t]))
    yy = np.arange(h, dtype=np.int32)[:, None]
    xx = np.arange(w, dtype=np.int32)[None, :]
    stripe = (((xx * 1280 // w) + 3 * (yy * 720 // h) + 2 * t) // 96) % 2
    img = np.empty((h, w, 3), np.uint8)
    img[..., 0] = 40 + 10 * stripe          # B
    img[..., 1] = 120 + 24 * stripe         # G
    img[..., 2] = 48 + 8 * stripe           # R
    img[: h // 8] = (90, 70, 60)            # stands
    Hm = camera(seed, t)
    sx, sy = w / 1280.0, h / 720.0
    s = np.linspace(0.0, 1.0, 1400)[:, None]
    for a, b in _SEGMENTS:
        pa = np.array(LANDMARKS[a][2:4]); pb = np.array(LANDMARKS[b][2:4])
        p = project(Hm, pa[None] * (1 - s) + pb[None] * s)
        _stamp(img, (p[:, 0] * sx).astype(np.int64), (p[:, 1] * sy).astype(np.int64), (235, 235, 235))
    th = np.linspace(0, 2 * np.pi, 1200)
    circ = np.stack([52.5 + 9.15 * np.cos(th), 34.0 + 9.15 * np.sin(th)], 1)
    p = project(Hm, circ)
    _stamp(img, (p[:, 0] * sx).astype(np.int64), (p[:, 1] * sy).astype(np.int64), (235, 235, 235))
    # players: world positions drift smoothly; drawn as upright rectangles with the foot on the ground point
    for k, x0, y0, x1, y1, ph_ in player_boxes(seed, t, h, w, _with_height=True):
        img[y0:y1, x0:x1] = _PLAYER_COLOURS[k]
        img[y0:min(y0 + max(3, ph_ // 6), y1), x0:x1] = (150, 170, 215)
    bpos = project(Hm, np.array([[52.5 + 20 * np.sin(0.02 * t), 34.0 + 12 * np.cos(0.017 * t)]]))[0]
    by, bx = np.ogrid[-4:5, -4:5]
    disc = (by * by + bx * bx) <= 12
    cy_, cx_ = int(bpos[1] * sy), int(bpos[0] * sx)
    if 5 <= cy_ < h - 5 and 5 <= cx_ < w - 5:
        img[cy_ - 4:cy_ + 5, cx_ - 4:cx_ + 5][disc] = (250, 250, 250)
    noise = r.integers(-6, 7, (h, w, 3), dtype=np.int16)
    return np.clip(img.astype(np.int16) + noise, 0, 255).astype(np.uint8)


def noise_frame(seed, h=720, w=1280):
    return np.random.Generator(np.random.PCG64([seed, 9])).integers(0, 256, (h, w, 3), dtype=np.uint8)


def clip(seed, n, h=720, w=1280, distinct=None):
    """[n,h,w,3] uint8 clip.  ``distinct`` < n tiles that many generated frames (bench memory/time bound;
    every frame still goes through the whole path)."""
    d = n if distinct is None else min(n, distinct)
    base = np.stack([frame(seed, t, h, w) for t in range(d)])
    if d == n:
        return base
    reps = (n + d - 1) // d
    return np.concatenate([base] * reps)[:n]

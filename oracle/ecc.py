"""ORACLE — test infrastructure only (see eo_prims.c header).

Camera-motion compensation of the tracker the reference constructs (eagle/models/coordinate_model.py:66-72: ``BotSort(...)`` with boxmot's
defaults, i.e. ``cmc_method="ecc"``; the frame reaches it at cm.py:577).  boxmot 15.0.2 (uv.lock:98-99) and cv2 are absent from
/root/reference and from this image: restated from the published sources — boxmot's ``ECC`` wrapper (gray -> ``cv2.resize(fx=fy=0.15)`` ->
``cv2.findTransformECC(prev, cur, eye(2, 3), MOTION_EUCLIDEAN, (EPS | COUNT, 100, 1e-5), None, 1)``, translation divided by the scale, identity
and the OLD template kept when the call raises) and OpenCV's ``findTransformECC`` (Evangelidis & Psarakis 2008, forward additive ECC;
video/src/ecc.cpp) with the primitives it calls: ``filter2D`` with (-0.5, 0, 0.5) and reflect-101 borders, ``warpAffine`` with
``WARP_INVERSE_MAP`` (10-bit fixed-point coordinates, 1/32-pixel bilinear table, constant border 0; nearest for the mask), ``meanStdDev`` /
``Mat::dot`` (double accumulators over float32 elements), the closed-form 3 x 3 ``invert``.  PARITY UNPINNED (no golden vector of cv2 exists
here); last-bit questions that the publication leaves open (FMA contraction inside cv2's SIMD paths, the accumulation width of the 3 x 3
matrix products) are decided here the plain way: separate float32 operations, float64 products rounded once.
"""
import numpy as np

from . import prims as P

AB_BITS, INTER_BITS = 10, 5
AB_SCALE, TAB = 1 << AB_BITS, 1 << INTER_BITS
F32 = np.float32


def preprocess(frame_bgr, scale=0.15):
    """boxmot BaseCMC.preprocess: cvtColor(BGR2GRAY) then cv2.resize(img, (0, 0), fx=scale, fy=scale, INTER_LINEAR) (dsize = round(size * scale))"""
    g = P.bgr2gray(frame_bgr)
    return P.resize_linear_u8c3_fxfy(np.repeat(g[:, :, None], 3, axis=2), scale, scale)[:, :, 0].copy()


def gradients(img):
    """filter2D(imageFloat, -1, [-0.5, 0, 0.5]) and its transpose, BORDER_REFLECT_101 (exact in float32: multiples of 0.5)"""
    f = img.astype(F32)
    px = np.pad(f, ((0, 0), (1, 1)), mode="reflect"); py = np.pad(f, ((1, 1), (0, 0)), mode="reflect")
    return F32(0.5) * (px[:, 2:] - px[:, :-2]), F32(0.5) * (py[2:, :] - py[:-2, :])


def _fixed_coords(M, h, w, nearest):
    M = np.asarray(M, np.float64)
    x = np.arange(w, dtype=np.float64); y = np.arange(h, dtype=np.float64)
    ad = np.rint(M[0, 0] * x * AB_SCALE).astype(np.int64); bd = np.rint(M[1, 0] * x * AB_SCALE).astype(np.int64)
    rd = AB_SCALE // 2 if nearest else AB_SCALE // TAB // 2
    X0 = np.rint((M[0, 1] * y + M[0, 2]) * AB_SCALE).astype(np.int64) + rd
    Y0 = np.rint((M[1, 1] * y + M[1, 2]) * AB_SCALE).astype(np.int64) + rd
    return X0[:, None] + ad[None, :], Y0[:, None] + bd[None, :]


def warp_linear(src, M, h, w):
    """warpAffine(src float32, M, (w, h), INTER_LINEAR | WARP_INVERSE_MAP), constant border 0"""
    X, Y = _fixed_coords(M, h, w, False)
    X >>= AB_BITS - INTER_BITS; Y >>= AB_BITS - INTER_BITS
    sx, sy, ax, ay = X >> INTER_BITS, Y >> INTER_BITS, X & (TAB - 1), Y & (TAB - 1)
    fx = ax.astype(F32) * F32(1.0 / TAB); fy = ay.astype(F32) * F32(1.0 / TAB)
    wx0, wy0 = F32(1) - fx, F32(1) - fy
    sh, sw = src.shape

    def tap(yy, xx):
        ok = (xx >= 0) & (xx < sw) & (yy >= 0) & (yy < sh)
        return np.where(ok, src[np.clip(yy, 0, sh - 1), np.clip(xx, 0, sw - 1)], F32(0))

    return ((tap(sy, sx) * (wy0 * wx0) + tap(sy, sx + 1) * (wy0 * fx)) + tap(sy + 1, sx) * (fy * wx0)) + tap(sy + 1, sx + 1) * (fy * fx)


def warp_mask(M, h, w, sh, sw):
    """warpAffine(ones u8, M, (w, h), INTER_NEAREST | WARP_INVERSE_MAP): 1 where the nearest source pixel exists"""
    X, Y = _fixed_coords(M, h, w, True)
    sx, sy = X >> AB_BITS, Y >> AB_BITS
    return (sx >= 0) & (sx < sw) & (sy >= 0) & (sy < sh)


def _dot(a, b):
    return float(np.sum(a.astype(np.float64) * b.astype(np.float64)))


def _inv3(Hf):
    """cv::invert of a 3 x 3 float32 matrix: adjugate / determinant with double products, rounded to float32; all zeros if singular"""
    S = Hf.astype(np.float64)
    d = (S[0, 0] * (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) - S[0, 1] * (S[1, 0] * S[2, 2] - S[1, 2] * S[2, 0])
         + S[0, 2] * (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]))
    if d == 0.0:
        return np.zeros((3, 3), F32)
    d = 1.0 / d
    t = np.empty((3, 3), np.float64)
    t[0, 0] = (S[1, 1] * S[2, 2] - S[1, 2] * S[2, 1]) * d
    t[0, 1] = (S[0, 2] * S[2, 1] - S[0, 1] * S[2, 2]) * d
    t[0, 2] = (S[0, 1] * S[1, 2] - S[0, 2] * S[1, 1]) * d
    t[1, 0] = (S[1, 2] * S[2, 0] - S[1, 0] * S[2, 2]) * d
    t[1, 1] = (S[0, 0] * S[2, 2] - S[0, 2] * S[2, 0]) * d
    t[1, 2] = (S[0, 2] * S[1, 0] - S[0, 0] * S[1, 2]) * d
    t[2, 0] = (S[1, 0] * S[2, 1] - S[1, 1] * S[2, 0]) * d
    t[2, 1] = (S[0, 1] * S[2, 0] - S[0, 0] * S[2, 1]) * d
    t[2, 2] = (S[0, 0] * S[1, 1] - S[0, 1] * S[1, 0]) * d
    return t.astype(F32)


def _dot3(a, b):
    """3-element dot of float32 vectors in float64, left to right"""
    a = [float(v) for v in a]; b = [float(v) for v in b]
    return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]


def _mat3vec(A, v):
    return np.array([_dot3(A[r], v) for r in range(3)], np.float64).astype(F32)


def find_transform_ecc(template, image, max_iter=100, eps=1e-5, trace=None):
    """cv2.findTransformECC(template u8, image u8, eye(2, 3) float32, MOTION_EUCLIDEAN, (EPS | COUNT, max_iter, eps), None, 1).
    Returns (rho, warp float32 [2, 3], iterations) or None where OpenCV raises (NaN correlation, lambda_d <= 0)."""
    hs, ws = template.shape
    hd, wd = image.shape
    T = template.astype(F32); I = image.astype(F32)
    gx, gy = gradients(image)
    xg = np.arange(ws, dtype=F32)[None, :].repeat(hs, 0); yg = np.arange(hs, dtype=F32)[:, None].repeat(ws, 1)
    M = np.array([[1, 0, 0], [0, 1, 0]], F32)
    rho, last_rho = -1.0, -eps
    it = 0
    while it < max_iter and abs(rho - last_rho) >= eps:
        it += 1
        Iw = warp_linear(I, M, hs, ws); Gx = warp_linear(gx, M, hs, ws); Gy = warp_linear(gy, M, hs, ws)
        mask = warp_mask(M, hs, ws, hd, wd)
        n = int(mask.sum())
        if n == 0:
            return None                                   # meanStdDev of nothing: 0 / 0 norms -> NaN rho
        im = Iw[mask].astype(np.float64); tm = T[mask].astype(np.float64)
        i_mean, t_mean = im.sum() / n, tm.sum() / n
        i_var = max((im * im).sum() / n - i_mean * i_mean, 0.0); t_var = max((tm * tm).sum() / n - t_mean * t_mean, 0.0)
        Izm = np.where(mask, Iw - F32(i_mean), Iw)        # subtract(..., mask): pixels outside the mask keep the warped value
        Tzm = np.where(mask, T - F32(t_mean), F32(0))
        t_norm = np.sqrt(n * t_var); i_norm = np.sqrt(n * i_var)
        h0, h1 = M[0, 0], M[1, 0]
        hat_x = -(xg * h1) - (yg * h0); hat_y = (xg * h0) - (yg * h1)
        J = [Gx * hat_x + Gy * hat_y, Gx, Gy]
        Hs = np.empty((3, 3), F32)
        for a in range(3):
            for b in range(a, 3):
                Hs[a, b] = Hs[b, a] = F32(_dot(J[a], J[b]))
        Hinv = _inv3(Hs)
        corr = _dot(Tzm, Izm)
        last_rho = rho
        with np.errstate(all="ignore"):
            rho = corr / (i_norm * t_norm)
        if np.isnan(rho):
            return None
        ip = np.array([_dot(j, Izm) for j in J], F32); tp = np.array([_dot(j, Tzm) for j in J], F32)
        iph = _mat3vec(Hinv, ip)
        lam_n = i_norm * i_norm - _dot3(ip, iph)
        lam_d = corr - _dot3(tp, iph)
        if lam_d <= 0.0:
            return None
        lam = lam_n / lam_d
        err = F32(lam) * Tzm - Izm
        ep = np.array([_dot(j, err) for j in J], F32)
        dp = _mat3vec(Hinv, ep)
        theta = float(dp[0]) + float(np.arcsin(np.float64(M[1, 0])))
        M[0, 2] += dp[1]; M[1, 2] += dp[2]
        M[0, 0] = M[1, 1] = F32(np.cos(theta)); M[1, 0] = F32(np.sin(theta)); M[0, 1] = -M[1, 0]
        if trace is not None:
            trace.append((rho, M.copy()))
    return rho, M, it


class ECC:
    """boxmot's ECC camera-motion estimator (stateful: keeps the previous pre-processed frame; a failed alignment returns the identity
    and keeps the OLD template)."""

    def __init__(self, scale=0.15, max_iter=100, eps=1e-5):
        self.scale, self.max_iter, self.eps = scale, max_iter, eps
        self.prev = None

    def apply(self, frame_bgr):
        ident = np.array([[1, 0, 0], [0, 1, 0]], F32)
        img = preprocess(frame_bgr, self.scale)
        if self.prev is None:
            self.prev = img
            return ident
        r = find_transform_ecc(self.prev, img, self.max_iter, self.eps)
        if r is None:
            return ident
        W = r[1].copy()
        W[0, 2] = F32(np.float64(W[0, 2]) / self.scale)     # warp_matrix[0, 2] /= self.scale (double quotient stored to float32)
        W[1, 2] = F32(np.float64(W[1, 2]) / self.scale)
        self.prev = img
        return W


def clip_motion(frames, scale=0.15):
    """[n, 6] float64 rows (2 x 3 row-major warp of frame i-1 -> i; row 0 = identity): what BotSort's ``self.cmc.apply(img, dets)`` returns per frame"""
    e = ECC(scale)
    return np.stack([e.apply(f).astype(np.float64).reshape(6) for f in frames]) if len(frames) else np.zeros((0, 6))

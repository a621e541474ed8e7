"""ctypes bindings of oracle/eo_prims.c  (ORACLE — test infrastructure only; see eo_prims.c header)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libeagle_oracle.so")

ACT_NONE, ACT_RELU, ACT_SILU = 0, 1, 2


def build(force=False):
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < max(os.path.getmtime(os.path.join(_HERE, f)) for f in ("eo_prims.c", "eo_flow.c")):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        # libgomp sizes its team from the logical CPU count; on the 256-thread GPU hosts (where far fewer cores are granted to the job) a
        # full-size team makes one HRNet forward take 49 s instead of 0.7 s.  Must be set before the library (and libgomp) is loaded.
        os.environ.setdefault("OMP_NUM_THREADS", str(max(1, min(16, len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else 16))))
        _lib = C.CDLL(_SO)
    return _lib


def _p(a, t=C.c_float):
    return a.ctypes.data_as(C.POINTER(t)) if a is not None else None


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def conv2d(x, w, b, stride=1, pad=None, pre=ACT_NONE, r1=None, r2=None, post=ACT_NONE, f16_out=False):
    """x [N,H,W,Cin], w [KS,KS,Cin,Cout] (BN folded), b [Cout] -> [N,Ho,Wo,Cout]"""
    x = _f32(x); w = _f32(w); b = _f32(b)
    N, H, W, Cin = x.shape
    KS, _, Cin2, Cout = w.shape
    assert Cin == Cin2, (x.shape, w.shape)
    if pad is None:
        pad = KS // 2
    Ho = (H + 2 * pad - KS) // stride + 1
    Wo = (W + 2 * pad - KS) // stride + 1
    y = np.empty((N, Ho, Wo, Cout), np.float32)
    if r1 is not None:
        r1 = _f32(r1); assert r1.shape == y.shape
    if r2 is not None:
        r2 = _f32(r2); assert r2.shape == y.shape
    lib().eo_conv2d_nhwc(_p(x), N, H, W, Cin, _p(w), _p(b), Cout, KS, stride, pad, Ho, Wo, _p(y), pre,
                         _p(r1), _p(r2), post, int(f16_out))
    return y


def upsample_bilinear_ac(x, H, W):
    x = _f32(x)
    N, h, w, Cc = x.shape
    y = np.empty((N, H, W, Cc), np.float32)
    lib().eo_upsample_bilinear_ac(_p(x), N, h, w, Cc, H, W, _p(y))
    return y


def sigmoid(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().eo_sigmoid_array(_p(x), _p(y), C.c_int64(x.size))
    return y


def expf(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().eo_exp_array(_p(x), _p(y), C.c_int64(x.size))
    return y


def round_f16(x):
    x = _f32(x)
    y = np.empty_like(x)
    lib().eo_round_f16_array(_p(x), _p(y), C.c_int64(x.size))
    return y


def heatmap_argmax(logits, n_real):
    """logits [H,W,Cs] -> (idx int32[n_real] flat row-major, score float32[n_real])"""
    logits = _f32(logits)
    H, W, Cs = logits.shape
    idx = np.empty(n_real, np.int32)
    sc = np.empty(n_real, np.float32)
    lib().eo_heatmap_argmax(_p(logits), H * W, Cs, n_real, _p(idx, C.c_int32), _p(sc))
    return idx, sc


def resize_linear_u8c3(src, dh, dw):
    src = np.ascontiguousarray(src, dtype=np.uint8)
    sh, sw, c = src.shape
    assert c == 3
    dst = np.empty((dh, dw, 3), np.uint8)
    lib().eo_resize_linear_u8c3(_p(src, C.c_uint8), sh, sw, C.c_int64(sw * 3), _p(dst, C.c_uint8), dh, dw)
    return dst


def resize_linear_u8c3_fxfy(src, fx, fy):
    """cv2.resize(src, (0, 0), fx=fx, fy=fy, interpolation=INTER_LINEAR): dsize = round(size * f), coordinates mapped with 1 / f"""
    src = np.ascontiguousarray(src, dtype=np.uint8)
    sh, sw, c = src.shape
    assert c == 3
    dh, dw = int(np.rint(sh * fy)), int(np.rint(sw * fx))
    dst = np.empty((dh, dw, 3), np.uint8)
    lib().eo_resize_linear_u8c3_fxfy(_p(src, C.c_uint8), sh, sw, C.c_int64(sw * 3), _p(dst, C.c_uint8), dh, dw, C.c_double(fx), C.c_double(fy))
    return dst


def yolo_decode_level(box, cls, nc, stride):
    """box [gh,gw,>=64] logits, cls [gh,gw,>=nc] logits -> [gh*gw, 4+nc]"""
    box = _f32(box); cls = _f32(cls)
    gh, gw, bcs = box.shape
    out = np.empty((gh * gw, 4 + nc), np.float32)
    lib().eo_yolo_decode_level(_p(box), bcs, _p(cls), cls.shape[2], nc, gh, gw, C.c_float(stride), _p(out))
    return out


def fit_line_l2(pts):
    pts = _f32(pts).reshape(-1, 2)
    line = np.empty(4, np.float32)
    lib().eo_fit_line_l2(_p(pts), pts.shape[0], _p(line))
    return line


def dlt_homography(src, dst):
    src = np.ascontiguousarray(src, np.float64).reshape(-1, 2)
    dst = np.ascontiguousarray(dst, np.float64).reshape(-1, 2)
    H = np.empty(9, np.float64)
    ok = lib().eo_dlt_homography(_p(src, C.c_double), _p(dst, C.c_double), None, src.shape[0], _p(H, C.c_double))
    return H.reshape(3, 3) if ok else None


def find_homography(src, dst, method=8, thresh=5.0, max_iters=2000, confidence=0.995, refine_iters=10, cv_solver=False):
    """cv2.findHomography(src, dst, method, thresh): method 8 = cv2.RANSAC, 16 = cv2.RHO (PROSAC + SPRT, eo_find_homography_rho), 4 = cv2.LMEDS.  cv_solver=True: OpenCV's own minimal solver
    and Jacobi (the second CPU mode of eo_prims.c) instead of the production deviations.  -> (H float64 3x3, mask uint8 [n,1]) | (None, None)"""
    src = _f32(src).reshape(-1, 2); dst = _f32(dst).reshape(-1, 2)
    n = src.shape[0]
    H = np.zeros(9, np.float64); mask = np.zeros(max(n, 1), np.uint8)
    f = lib().eo_find_homography_ex
    f.restype = C.c_int
    ok = f(_p(src), _p(dst), n, int(method), C.c_double(thresh), int(max_iters), C.c_double(confidence), int(refine_iters), int(bool(cv_solver)),
           _p(H, C.c_double), mask.ctypes.data_as(C.POINTER(C.c_uint8)))
    return (H.reshape(3, 3), mask[:n].reshape(-1, 1)) if ok else (None, None)


def find_homography_ransac(src, dst, thresh=5.0, max_iters=2000, confidence=0.995, refine_iters=10):
    """cv2.findHomography(src, dst, cv2.RANSAC, thresh) restatement -> (H float64 3x3 | None, mask u8 [n] | None)"""
    src = _f32(src).reshape(-1, 2)
    dst = _f32(dst).reshape(-1, 2)
    n = src.shape[0]
    H = np.empty(9, np.float64)
    mask = np.zeros(max(n, 1), np.uint8)
    ok = lib().eo_find_homography_ransac(_p(src), _p(dst), n, C.c_double(thresh), max_iters, C.c_double(confidence),
                                         refine_iters, _p(H, C.c_double), _p(mask, C.c_uint8))
    if not ok:
        return None, None
    return H.reshape(3, 3), mask[:n]


def perspective_transform(pts, H):
    pts = _f32(pts).reshape(-1, 2)
    H = np.ascontiguousarray(H, np.float64).reshape(9)
    out = np.empty_like(pts)
    lib().eo_perspective_transform(_p(pts), pts.shape[0], _p(H, C.c_double), _p(out))
    return out


# ---- optical-flow row (eo_flow.c) -------------------------------------------------------------------------------
def bgr2gray(bgr):
    """cv2.cvtColor(frame, cv2.COLOR_BGR2GRAY) (cm.py:280)"""
    bgr = np.ascontiguousarray(bgr, np.uint8)
    out = np.empty(bgr.shape[:2], np.uint8)
    lib().eo_bgr2gray(_p(bgr, C.c_uint8), bgr.shape[0], bgr.shape[1], _p(out, C.c_uint8))
    return out


def bgr2hsv(bgr):
    """cv2.cvtColor(grid, cv2.COLOR_BGR2HSV) on uint8 (cm.py:459,469,538,545)"""
    bgr = np.ascontiguousarray(bgr, np.uint8)
    out = np.empty(bgr.shape, np.uint8)
    lib().eo_bgr2hsv(_p(bgr, C.c_uint8), C.c_long(bgr.size // 3), _p(out, C.c_uint8))
    return out


def pyrdown(gray):
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    out = np.empty(((h + 1) // 2, (w + 1) // 2), np.uint8)
    lib().eo_pyrdown(_p(gray, C.c_uint8), h, w, _p(out, C.c_uint8))
    return out


def calc_optical_flow_pyr_lk(prev_gray, next_gray, prev_pts, max_level=2, max_count=10, epsilon=0.03):
    """cv2.calcOpticalFlowPyrLK(prev_gray, curr_gray, prev_points, None, winSize=(15,15), maxLevel=2,
    criteria=(EPS|COUNT, 10, 0.03)) (cm.py:65,434) -> (next_pts (n,2) f32, status (n,1) u8)"""
    prev_gray = np.ascontiguousarray(prev_gray, np.uint8); next_gray = np.ascontiguousarray(next_gray, np.uint8)
    assert prev_gray.shape == next_gray.shape and prev_gray.ndim == 2
    pts = _f32(prev_pts).reshape(-1, 2)
    n = len(pts)
    nxt = np.zeros((n, 2), np.float32); st = np.zeros((n, 1), np.uint8)
    lib().eo_calc_optical_flow_pyr_lk(_p(prev_gray, C.c_uint8), _p(next_gray, C.c_uint8), prev_gray.shape[0], prev_gray.shape[1],
                                      _p(pts), n, max_level, max_count, C.c_double(epsilon), _p(nxt), _p(st, C.c_uint8))
    return nxt, st

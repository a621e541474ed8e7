"""ctypes binding of ``libeagle_hip.so`` (C ABI: ``include/eagle.h``).

There is deliberately **no fallback**: if the shared library is missing or a call fails, an exception is raised.
torch is not imported here — the library talks to HIP directly."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EAGLE_HIP_LIB") or os.path.join(_HERE, "libeagle_hip.so")   # override: developer A/B builds only

MAX_DET, N_LANDMARKS, MAX_KP = 300, 57, 87
PREC_F16, PREC_F32, PREC_F32S = 0, 1, 2
PRECISIONS = {"f16": PREC_F16, "f32": PREC_F32, "f32s": PREC_F32S}      # include/eagle.h EAGLE_PREC_*
DET_VARIANTS = {"n": 0, "s": 1, "m": 2, "l": 3, "x": 4}
AUTO, SMALL_BATCH = -1, 32                                             # include/eagle.h EAGLE_AUTO / EAGLE_SMALL_BATCH (use_graph: 0 off, 1 every step, 2 inside calls of >= 3 steps; multi_stream)
LETTERBOX = {"rect": 0, "square": 1}                                    # include/eagle.h EAGLE_LETTERBOX_*: ultralytics LetterBox auto=True (the .pt predictor) / auto=False (the exported ONNX detector of cm.py:54-55)
DET_PREC_AUTO = -1                                                     # include/eagle.h EAGLE_DET_PREC_AUTO
DET_PREC_MIXED = 4                                                     # include/eagle.h EAGLE_DET_PREC_MIXED (split trunk, exact last C2f per level + Detect)


class EagleError(RuntimeError):
    pass


class EagleRangeError(EagleError):
    """EAGLE_E_RANGE: the split-precision family clipped an activation at +-4094 (include/eagle.h).  The records were written and are attached as
    ``.records`` (``pad[1]`` marks the frames); they are not fp32-grade."""
    records = None


class EagleConfig(C.Structure):
    _fields_ = [("device", C.c_int32), ("frame_h", C.c_int32), ("frame_w", C.c_int32), ("det_variant", C.c_int32),
                ("det_imgsz", C.c_int32), ("batch", C.c_int32), ("precision", C.c_int32),
                ("keypoint_conf", C.c_double), ("detector_conf", C.c_double), ("ransac_thresh", C.c_double),
                ("detector_floor", C.c_float), ("nms_iou", C.c_float),
                ("ransac_max_iters", C.c_int32), ("lm_iters", C.c_int32), ("use_graph", C.c_int32), ("det_precision", C.c_int32),
                ("allow_saturation", C.c_int32), ("multi_stream", C.c_int32), ("letterbox", C.c_int32), ("reserved", C.c_int32 * 3)]


class EagleTimings(C.Structure):
    _fields_ = [("total_ms", C.c_float), ("conv_ms", C.c_float), ("n_launches", C.c_int32),
                ("n_conv_launches", C.c_int32), ("conv_flop", C.c_double), ("sat_events", C.c_int32), ("sat_frames", C.c_int32),
                ("graph_captures", C.c_int32), ("graph_skipped", C.c_int32), ("reserved", C.c_int32 * 4)]


class EagleTrackParams(C.Structure):
    _fields_ = [("track_high_thresh", C.c_float), ("track_low_thresh", C.c_float), ("new_track_thresh", C.c_float), ("match_thresh", C.c_float),
                ("track_buffer", C.c_int32), ("frame_rate", C.c_int32)]


class EagleKernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 40), ("ms", C.c_float), ("launches", C.c_int32), ("bytes", C.c_double), ("flop", C.c_double)]


# numpy mirrors of the record structs (C layout, natural alignment — checked against sizeof in tests)
DET_DTYPE = np.dtype([("x1", "<f4"), ("y1", "<f4"), ("x2", "<f4"), ("y2", "<f4"), ("conf", "<f4"), ("cls", "<i4"),
                      ("id", "<i4"), ("bx1", "<i4"), ("by1", "<i4"), ("bx2", "<i4"), ("by2", "<i4"),
                      ("foot_x", "<i4"), ("foot_y", "<i4"), ("pitch_xf", "<f4"), ("pitch_yf", "<f4"),
                      ("pitch_x", "<i4"), ("pitch_y", "<i4"), ("reported", "u1"), ("in_bounds", "u1"), ("pad", "u1", 2)],
                     align=True)
KP_DTYPE = np.dtype([("label", "<i4"), ("x", "<i4"), ("y", "<i4"), ("score", "<f4"), ("synthesized", "u1"),
                     ("on_plane", "u1"), ("inlier", "u1"), ("pad", "u1")], align=True)
RESULT_DTYPE = np.dtype([("n_det", "<i4"), ("n_kp", "<i4"), ("n_candidates", "<i4"), ("H_valid", "u1"),
                         ("bounds_valid", "u1"), ("pad", "u1", 2), ("H", "<f8", 9), ("bounds", "<f8", 4),
                         ("hm_idx", "<i4", N_LANDMARKS), ("hm_score", "<f4", N_LANDMARKS),
                         ("kp", KP_DTYPE, MAX_KP), ("det", DET_DTYPE, MAX_DET)], align=True)

_lib = None


def require_torch_first():
    """One process must hold ONE ROCm runtime.  PyTorch-ROCm wheels bundle their own libamdhip64 / libhsa-runtime64 / librccl (SONAMEs identical
    to /opt/rocm's): when torch is imported BEFORE this library is loaded, the dynamic loader binds libeagle_hip.so and the dlopen'ed RCCL to
    torch's copies by SONAME — one runtime.  The other order maps /opt/rocm's runtime first and torch's second copy next to it (torch asks for
    "libamdhip64.so", which matches no SONAME): two HSA runtimes in one process, which is what aborted at interpreter exit in round 3
    (DESIGN.md §8).  Called by every code path of this package that imports torch after the library may have been loaded.  The gate is the
    dlopen itself (``load()``: it is libeagle_hip.so's NEEDED entry that maps /opt/rocm's libamdhip64), not the first handle."""
    import sys
    if "torch" not in sys.modules and _lib is not None:
        raise EagleError("torch must be imported before eagle_amd.lib loads libeagle_hip.so in a process that uses both "
                         "(two ROCm runtimes would be mapped: torch's bundled one and /opt/rocm's); import torch first, or keep the process torch-free")


def load():
    """Load the shared library and declare the prototypes.  Raises EagleError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EagleError(f"{LIB_PATH} not found: run `python -c 'import __graft_entry__ as g; g.build()'` "
                         "(there is no CPU fallback for the product path)")
    L = C.CDLL(LIB_PATH)
    vp, i32, i64 = C.c_void_p, C.c_int, C.c_int64
    fp, dp, u8p = C.POINTER(C.c_float), C.POINTER(C.c_double), C.POINTER(C.c_uint8)
    L.eagle_abi_sizes.argtypes = [C.POINTER(C.c_int32)]
    L.eagle_default_config.argtypes = [C.POINTER(EagleConfig)]
    L.eagle_create.argtypes = [C.POINTER(EagleConfig), C.POINTER(vp)]
    L.eagle_destroy.argtypes = [vp]
    L.eagle_destroy.restype = None
    L.eagle_last_error.argtypes = [vp]
    L.eagle_last_error.restype = C.c_char_p
    L.eagle_get_config.argtypes = [vp, C.POINTER(EagleConfig)]
    L.eagle_resolve_config.argtypes = [C.POINTER(EagleConfig)]
    L.eagle_load_weights.argtypes = [vp, C.c_char_p, fp, C.POINTER(i64), i32]
    L.eagle_finalize_weights.argtypes = [vp]
    L.eagle_process_frames.argtypes = [vp, u8p, i32, i64, i64, vp]
    L.eagle_process_device_frames.argtypes = [vp, vp, i32, vp]
    L.eagle_host_alloc.argtypes = [vp, i64, C.POINTER(vp)]
    L.eagle_host_free.argtypes = [vp, vp]
    L.eagle_device_alloc.argtypes = [vp, i64, C.POINTER(vp)]
    L.eagle_device_free.argtypes = [vp, vp]
    L.eagle_device_upload.argtypes = [vp, vp, vp, i64]
    L.eagle_reproject.argtypes = [vp, vp, i32, dp, u8p]
    L.eagle_clip_open.argtypes = [vp, vp, i32]
    L.eagle_clip_close.argtypes = [vp]
    L.eagle_clip_detect_keypoints.argtypes = [vp, i32, i32, i32]
    L.eagle_clip_get_keypoints.argtypes = [vp, i32, vp, C.POINTER(i32)]
    L.eagle_clip_set_keypoints.argtypes = [vp, i32, vp, i32]
    L.eagle_clip_flow.argtypes = [vp, i32, i32, i32, vp, i32, vp, C.POINTER(i32), fp, u8p]
    L.eagle_clip_run.argtypes = [vp, i32, i32, i32, i32, i32, i32, C.POINTER(i32)]
    L.eagle_clip_detect_objects.argtypes = [vp, i32, i32]
    L.eagle_clip_fetch.argtypes = [vp, vp]
    L.eagle_comm_id.argtypes = [vp]
    L.eagle_comm_init.argtypes = [vp, i32, i32, vp]
    L.eagle_gather.argtypes = [vp, vp, i32, vp]
    L.eagle_set_profiling.argtypes = [vp, i32]
    L.eagle_get_timings.argtypes = [vp, C.POINTER(EagleTimings)]
    L.eagle_get_kernel_times.argtypes = [vp, C.POINTER(EagleKernelTime), i32, C.POINTER(i32)]
    L.eagle_op_conv2d.argtypes = [i32, i32, fp, i32, i32, i32, i32, fp, fp, i32, i32, i32, i32, fp, fp, i32, fp]
    L.eagle_op_bottleneck.argtypes = [i32, fp, i32, i32, i32, i32, fp, fp, fp, fp, fp, fp, fp, fp, i32, fp, fp, fp]
    L.eagle_op_fuse_sum.argtypes = [i32, i32, fp, i32, i32, i32, i32, i32, C.POINTER(fp), C.POINTER(i32), C.POINTER(i32), i32, fp]
    L.eagle_op_preprocess.argtypes = [i32, i32, u8p, i32, i32, i32, i32, fp, fp, C.POINTER(i32)]
    L.eagle_op_preprocess_lb.argtypes = [i32, i32, u8p, i32, i32, i32, i32, i32, fp, fp, C.POINTER(i32)]
    L.eagle_op_find_homography.argtypes = [i32, fp, fp, i32, C.c_double, i32, i32, dp, u8p, C.POINTER(i32)]
    L.eagle_debug.argtypes = [C.c_char_p, i64, vp, i64]
    L.eagle_team_colors.argtypes = [vp, vp, i32, vp, i32, vp]
    L.eagle_track_open.argtypes = [vp, C.POINTER(EagleTrackParams)]
    L.eagle_track_frames.argtypes = [vp, vp, i32]
    L.eagle_track_frames_cmc.argtypes = [vp, vp, i32, C.POINTER(C.c_double)]
    L.eagle_clip_motion.argtypes = [vp, i32, i32, C.POINTER(C.c_double)]
    L.eagle_clip_motion_ecc.argtypes = [vp, i32, i32, i32, C.POINTER(C.c_double), C.POINTER(i32)]
    L.eagle_reid_features.argtypes = [vp, vp, i32, vp, i32, fp]
    L.eagle_track_frames_reid.argtypes = [vp, vp, i32, C.POINTER(C.c_double), fp, C.POINTER(i32), C.POINTER(i32)]
    _lib = L
    return L


EXPORTS = ["eagle_abi_sizes", "eagle_default_config", "eagle_create", "eagle_destroy", "eagle_last_error", "eagle_get_config", "eagle_resolve_config", "eagle_load_weights",
           "eagle_finalize_weights", "eagle_process_frames", "eagle_process_device_frames", "eagle_device_alloc",
           "eagle_device_free", "eagle_device_upload", "eagle_host_alloc", "eagle_host_free", "eagle_reproject", "eagle_comm_id", "eagle_comm_init", "eagle_gather",
           "eagle_set_profiling", "eagle_get_timings", "eagle_get_kernel_times", "eagle_op_conv2d", "eagle_op_bottleneck", "eagle_op_fuse_sum", "eagle_op_preprocess", "eagle_op_preprocess_lb",
           "eagle_op_find_homography", "eagle_clip_open", "eagle_clip_close", "eagle_clip_detect_objects", "eagle_clip_detect_keypoints", "eagle_clip_get_keypoints",
           "eagle_clip_set_keypoints", "eagle_clip_flow", "eagle_clip_run", "eagle_clip_fetch", "eagle_debug", "eagle_track_open", "eagle_track_frames", "eagle_track_frames_cmc", "eagle_clip_motion_ecc", "eagle_clip_motion", "eagle_team_colors",
           "eagle_reid_features", "eagle_track_frames_reid"]

FLOWKP_DTYPE = np.dtype([("label", "<i4"), ("x", "<i4"), ("y", "<i4"), ("score", "<f4")], align=True)
E_REFERENCE_RAISES = -7
E_RANGE = -8


def debug(key, value=0, out=None):
    """Developer diagnostics (include/eagle.h, eagle_debug)."""
    rc = load().eagle_debug(key.encode(), int(value), None if out is None else out.ctypes.data_as(C.c_void_p), 0 if out is None else out.nbytes)
    if rc:
        raise EagleError(f"eagle_debug({key}) failed ({rc})")
    return out


def resolve_config(cfg):
    """A copy of cfg with the "auto" fields resolved the way eagle_create will (no GPU needed)."""
    out = EagleConfig.from_buffer_copy(cfg)
    if load().eagle_resolve_config(C.byref(out)):
        raise EagleError("eagle_resolve_config failed")
    return out


def abi_sizes():
    o = (C.c_int32 * 4)()
    load().eagle_abi_sizes(o)
    return list(o)


def _fp(a):
    return a.ctypes.data_as(C.POINTER(C.c_float)) if a is not None else None


def default_config(**kw):
    """eagle_default_config + overrides: a thin pass-through.  The library's default ``det_precision`` is EAGLE_DET_PREC_AUTO (-1), which
    eagle_create resolves from ``precision``: the exact fp32 family next to split-family key-points, otherwise the key-point family
    (``Handle.cfg`` holds the resolved value)."""
    cfg = EagleConfig()
    load().eagle_default_config(C.byref(cfg))
    for k, v in kw.items():
        if k == "det_variant" and isinstance(v, str):
            v = DET_VARIANTS[v]
        if k == "letterbox" and isinstance(v, str):
            v = LETTERBOX[v]
        if not hasattr(cfg, k):
            raise TypeError(f"unknown config field {k}")
        setattr(cfg, k, v)
    return cfg


class Handle:
    """One library handle == one GPU worker (not thread-safe)."""

    def __init__(self, cfg=None, **kw):
        self.L = load()
        self.cfg = cfg or default_config(**kw)
        self._h = C.c_void_p()
        rc = self.L.eagle_create(C.byref(self.cfg), C.byref(self._h))
        if rc:
            raise EagleError(f"eagle_create failed ({rc}): {self.L.eagle_last_error(None).decode()}")
        resolved = EagleConfig()
        self._check(self.L.eagle_get_config(self._h, C.byref(resolved)), "get_config")
        self.cfg = resolved                                 # det_precision as the library resolved it (EAGLE_DET_PREC_AUTO -> a family)

    def _check(self, rc, what):
        if rc == E_RANGE:
            raise EagleRangeError(f"{what}: {self.L.eagle_last_error(self._h).decode()}")
        if rc:
            raise EagleError(f"{what} failed ({rc}): {self.L.eagle_last_error(self._h).decode()}")

    def _check_records(self, rc, what, out):
        try:
            self._check(rc, what)
        except EagleRangeError as e:
            e.records = out
            raise

    def close(self):
        if self._h:
            self.L.eagle_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def load_weight(self, name, arr):
        arr = np.ascontiguousarray(arr, np.float32)
        shape = (C.c_int64 * max(arr.ndim, 1))(*arr.shape)
        self._check(self.L.eagle_load_weights(self._h, name.encode(), _fp(arr), shape, arr.ndim), f"load_weights({name})")

    def finalize_weights(self):
        self._check(self.L.eagle_finalize_weights(self._h), "finalize_weights")

    def process(self, frames, out=None, strides=None):
        """frames: uint8 [n,h,w,3] BGR (host).  -> structured array [n] of RESULT_DTYPE (``out``: optional preallocated result array).

        The array is handed over AS THE VIEW IT IS (cm.py:568 passes whatever view the caller holds): a crop of a wider surface or a decoder's
        padded plane goes to eagle_process_frames with its own frame / row strides, no host-side copy.  Only views whose pixels are not packed
        BGR triples, whose strides are negative or whose frames overlap are made contiguous first.  ``strides=(frame_stride, row_stride)`` in
        bytes overrides what is passed (tests of the boundary's argument checks)."""
        frames = np.asarray(frames)
        if frames.dtype != np.uint8:
            frames = frames.astype(np.uint8)
        if frames.ndim == 3:
            frames = frames[None]
        n, h, w, c = frames.shape
        if (h, w, c) != (self.cfg.frame_h, self.cfg.frame_w, 3):
            raise EagleError(f"frame shape {(h, w, c)} does not match the handle ({self.cfg.frame_h}, {self.cfg.frame_w}, 3)")
        fs, rs, ps, cs = frames.strides
        if n == 1:
            fs = max(fs, rs * (h - 1) + 3 * w)              # numpy reports any stride for a length-1 axis
        if not (cs == 1 and ps == 3 and rs >= 3 * w and fs >= rs * (h - 1) + 3 * w):
            frames = np.ascontiguousarray(frames)
            fs, rs = h * w * 3, w * 3
        if strides is not None:
            fs, rs = strides
        if out is None:
            out = np.zeros(n, RESULT_DTYPE)
        assert out.dtype == RESULT_DTYPE and len(out) >= n and out.flags.c_contiguous
        self._check_records(self.L.eagle_process_frames(self._h, C.cast(frames.ctypes.data, C.POINTER(C.c_uint8)), n, int(fs), int(rs),
                                                        out.ctypes.data_as(C.c_void_p)), "process_frames", out)
        return out

    def host_frames(self, n):
        """uint8 [n,h,w,3] array in pinned host memory (eagle_host_alloc): what a decoder should write frames into so that
        eagle_process_frames can DMA them in place.  Release with host_free(array)."""
        shape = (n, self.cfg.frame_h, self.cfg.frame_w, 3)
        nbytes = int(np.prod(shape))
        p = C.c_void_p()
        self._check(self.L.eagle_host_alloc(self._h, nbytes, C.byref(p)), "host_alloc")
        buf = (C.c_uint8 * max(nbytes, 1)).from_address(p.value)
        a = np.frombuffer(buf, np.uint8, nbytes).reshape(shape)
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[a.ctypes.data] = p
        return a

    def host_buffer(self, nbytes):
        """uint8 [nbytes] in pinned host memory (eagle_host_alloc) for callers that lay frames out themselves (row / frame pitch).  host_free(array)."""
        p = C.c_void_p()
        self._check(self.L.eagle_host_alloc(self._h, int(nbytes), C.byref(p)), "host_alloc")
        a = np.frombuffer((C.c_uint8 * max(int(nbytes), 1)).from_address(p.value), np.uint8, int(nbytes))
        self._pinned = getattr(self, "_pinned", {})
        self._pinned[a.ctypes.data] = p
        return a

    def host_free(self, a):
        p = self._pinned.pop(a.ctypes.data)
        self._check(self.L.eagle_host_free(self._h, p), "host_free")

    def upload(self, frames):
        frames = np.ascontiguousarray(frames, np.uint8)
        d = C.c_void_p()
        self._check(self.L.eagle_device_alloc(self._h, frames.nbytes, C.byref(d)), "device_alloc")
        self._check(self.L.eagle_device_upload(self._h, d, frames.ctypes.data_as(C.c_void_p), frames.nbytes), "device_upload")
        return d

    def free(self, d):
        self._check(self.L.eagle_device_free(self._h, d), "device_free")

    def process_device(self, dptr, n, out=None):
        if out is None:
            out = np.zeros(n, RESULT_DTYPE)
        self._check_records(self.L.eagle_process_device_frames(self._h, dptr, n, out.ctypes.data_as(C.c_void_p)), "process_device_frames", out)
        return out

    def reproject(self, recs, Hs, flags):
        """In place: re-project foot points / boundaries of the flagged records with the given homographies (cadence mode)."""
        Hs = np.ascontiguousarray(Hs, np.float64).reshape(-1, 9)
        flags = np.ascontiguousarray(flags, np.uint8)
        assert recs.flags.c_contiguous and len(Hs) == len(flags) == len(recs)
        self._check(self.L.eagle_reproject(self._h, recs.ctypes.data_as(C.c_void_p), len(recs), Hs.ctypes.data_as(C.POINTER(C.c_double)),
                                           flags.ctypes.data_as(C.POINTER(C.c_uint8))), "reproject")
        return recs

    # --- clip session: optical-flow key-point cadence (include/eagle.h, eagle_clip_*) --------------------
    def clip_open(self, dptr, n):
        self._check(self.L.eagle_clip_open(self._h, dptr, n), "clip_open")

    def clip_close(self):
        self._check(self.L.eagle_clip_close(self._h), "clip_close")

    def clip_detect_objects(self, first, count):
        self._check(self.L.eagle_clip_detect_objects(self._h, first, count), "clip_detect_objects")

    def clip_detect_keypoints(self, first, stride=1, count=1):
        self._check(self.L.eagle_clip_detect_keypoints(self._h, first, stride, count), "clip_detect_keypoints")

    def clip_get_keypoints(self, frame):
        """mem[frame] as a FLOWKP_DTYPE array in dict order, or None when the frame has no entry."""
        buf = np.zeros(N_LANDMARKS, FLOWKP_DTYPE); n = C.c_int(0)
        self._check(self.L.eagle_clip_get_keypoints(self._h, frame, buf.ctypes.data_as(C.c_void_p), C.byref(n)), "clip_get_keypoints")
        return None if n.value < 0 else buf[: n.value].copy()

    def clip_set_keypoints(self, frame, kps):
        kps = np.ascontiguousarray(kps, FLOWKP_DTYPE)
        self._check(self.L.eagle_clip_set_keypoints(self._h, frame, kps.ctypes.data_as(C.c_void_p), len(kps)), "clip_set_keypoints")

    def clip_flow(self, src_frame, dst_frame, hue_frame, kps, raw=False):
        """calculate_optical_flow (cm.py:419-478) on resident frames -> filtered FLOWKP_DTYPE array (+ raw LK output)."""
        kps = np.ascontiguousarray(kps, FLOWKP_DTYPE)
        out = np.zeros(N_LANDMARKS, FLOWKP_DTYPE); n = C.c_int(0)
        nxt = np.zeros((max(len(kps), 1), 2), np.float32); st = np.zeros(max(len(kps), 1), np.uint8)
        self._check(self.L.eagle_clip_flow(self._h, src_frame, dst_frame, hue_frame, kps.ctypes.data_as(C.c_void_p), len(kps),
                                           out.ctypes.data_as(C.c_void_p), C.byref(n), _fp(nxt), st.ctypes.data_as(C.POINTER(C.c_uint8))), "clip_flow")
        return (out[: n.value].copy(), nxt[: len(kps)], st[: len(kps)]) if raw else out[: n.value].copy()

    def clip_run(self, first, last, keypoint_interval, homography_interval, calibration=False, wait=True):
        """Enqueues the loop body for frames [first, last) on the GPU; wait=True -> index of a frame that needs an on-demand
        detection, or -1 when every frame enqueued so far is done."""
        stalled = C.c_int(-1)
        rc = self.L.eagle_clip_run(self._h, first, last, keypoint_interval, homography_interval, int(calibration), int(wait), C.byref(stalled))
        if rc == E_REFERENCE_RAISES:
            raise IndexError(self.L.eagle_last_error(self._h).decode())
        self._check(rc, "clip_run")
        return stalled.value

    def clip_fetch(self, n):
        out = np.zeros(n, RESULT_DTYPE)
        self._check(self.L.eagle_clip_fetch(self._h, out.ctypes.data_as(C.c_void_p)), "clip_fetch")
        return out

    def team_colors(self, dptr, n_frames, crops):
        """crops: int32 [k,5] (frame, x1, y1, x2, y2) of a clip resident in HBM -> int32 [k,12] colour-range counts (include/eagle.h)."""
        crops = np.ascontiguousarray(crops, np.int32).reshape(-1, 5)
        out = np.zeros((len(crops), 12), np.int32)
        self._check(self.L.eagle_team_colors(self._h, dptr, n_frames, crops.ctypes.data_as(C.c_void_p), len(crops), out.ctypes.data_as(C.c_void_p)), "team_colors")
        return out

    # --- track identities (include/eagle.h, eagle_track_*) -------------------------------------------------
    def track_open(self, **params):
        p = None
        if params:
            d = dict(track_high_thresh=0.5, track_low_thresh=0.1, new_track_thresh=0.6, match_thresh=0.8, track_buffer=30, frame_rate=30)
            d.update(params)
            p = C.byref(EagleTrackParams(**d))
        self._check(self.L.eagle_track_open(self._h, p), "track_open")

    def reid_features(self, dptr, n_frames, crops):
        """crops: int32 [k,5] (frame, x1, y1, x2, y2) of a clip resident in HBM -> float32 [k, 512] OSNet-x0.25 embeddings (needs the reid.* weights)."""
        crops = np.ascontiguousarray(crops, np.int32).reshape(-1, 5)
        out = np.zeros((len(crops), 512), np.float32)
        self._check(self.L.eagle_reid_features(self._h, dptr, n_frames, crops.ctypes.data_as(C.c_void_p), len(crops), out.ctypes.data_as(C.POINTER(C.c_float))), "reid_features")
        return out

    def track_frames_reid(self, recs, feats, feat_det, feat_count, warps=None):
        """track_frames with appearance: record i owns feat_count[i] consecutive rows of feats; feat_det = their detection indices."""
        assert recs.dtype == RESULT_DTYPE and recs.flags.c_contiguous
        feats = np.ascontiguousarray(feats, np.float32).reshape(-1, 512)
        feat_det = np.ascontiguousarray(feat_det, np.int32); feat_count = np.ascontiguousarray(feat_count, np.int32)
        assert len(feat_count) == len(recs) and int(feat_count.sum()) == len(feats) == len(feat_det)
        w = None if warps is None else np.ascontiguousarray(warps, np.float64).reshape(len(recs), 6)
        self._check(self.L.eagle_track_frames_reid(self._h, recs.ctypes.data_as(C.c_void_p), len(recs), None if w is None else w.ctypes.data_as(C.POINTER(C.c_double)),
                                                   feats.ctypes.data_as(C.POINTER(C.c_float)), feat_det.ctypes.data_as(C.POINTER(C.c_int32)),
                                                   feat_count.ctypes.data_as(C.POINTER(C.c_int32))), "track_frames_reid")
        return recs

    def track_frames(self, recs, warps=None):
        """In place: the next records of the clip being tracked (frame order).  warps: optional [len(recs), 6] camera motions (clip_motion)."""
        assert recs.dtype == RESULT_DTYPE and recs.flags.c_contiguous
        w = None
        if warps is not None:
            w = np.ascontiguousarray(warps, np.float64).reshape(len(recs), 6)
        self._check(self.L.eagle_track_frames_cmc(self._h, recs.ctypes.data_as(C.c_void_p), len(recs),
                                                  w.ctypes.data_as(C.POINTER(C.c_double)) if w is not None else None), "track_frames")
        return recs

    def clip_motion(self, first, count):
        """[count, 6] float64: row-major 2 x 3 camera motion of frame first+i-1 -> first+i of the open clip session (identity for frame 0)."""
        w = np.zeros((max(count, 0), 6), np.float64)
        self._check(self.L.eagle_clip_motion(self._h, first, count, w.ctypes.data_as(C.POINTER(C.c_double))), "clip_motion")
        return w

    def clip_motion_ecc(self, first, count, carry=False, return_ok=False):
        """boxmot's default camera-motion estimator (ECC on the 0.15-scale gray frame): [count, 6] float64 warps like clip_motion; carry=True
        keeps the estimator's template across clips (the tracker's lifetime; track_open forgets it)."""
        w = np.zeros((max(count, 0), 6), np.float64)
        ok = np.zeros(max(count, 0), np.int32)
        self._check(self.L.eagle_clip_motion_ecc(self._h, first, count, int(bool(carry)), w.ctypes.data_as(C.POINTER(C.c_double)),
                                                 ok.ctypes.data_as(C.POINTER(C.c_int32))), "clip_motion_ecc")
        return (w, ok) if return_ok else w

    def set_profiling(self, on):
        self._check(self.L.eagle_set_profiling(self._h, int(on)), "set_profiling")

    def timings(self):
        t = EagleTimings()
        self._check(self.L.eagle_get_timings(self._h, C.byref(t)), "get_timings")
        return t

    def kernel_times(self):
        """[(name, total ms, launches, algorithmic bytes, flop)] since set_profiling(1): one row per non-convolution kernel and one per
        convolution layer shape ("conv 3x3/1 96->96 @68x120")."""
        buf = (EagleKernelTime * 256)(); n = C.c_int(0)
        self._check(self.L.eagle_get_kernel_times(self._h, buf, 256, C.byref(n)), "get_kernel_times")
        return [(buf[i].name.decode(), float(buf[i].ms), int(buf[i].launches), float(buf[i].bytes), float(buf[i].flop)) for i in range(min(n.value, 256))]

    # --- multi-GPU -------------------------------------------------------------------------------------
    def comm_init(self, rank, world, uid_bytes):
        buf = C.create_string_buffer(bytes(uid_bytes), 128)
        self._check(self.L.eagle_comm_init(self._h, rank, world, buf), "comm_init")

    def gather(self, local, world, out=None):
        local = np.ascontiguousarray(local)
        if out is None:
            out = np.zeros(len(local) * world, RESULT_DTYPE)
        assert out.dtype == RESULT_DTYPE and len(out) == len(local) * world and out.flags.c_contiguous
        self._check(self.L.eagle_gather(self._h, local.ctypes.data_as(C.c_void_p), len(local), out.ctypes.data_as(C.c_void_p)), "gather")
        return out


def comm_unique_id():
    buf = C.create_string_buffer(128)
    rc = load().eagle_comm_id(buf)
    if rc:
        raise EagleError(f"eagle_comm_id failed ({rc}): {load().eagle_last_error(None).decode()}")
    return buf.raw


# --- operator-level wrappers (parity tests) -------------------------------------------------------------------
def op_conv2d(x, w_hwio, bias, stride=1, pre=0, r1=None, r2=None, post=0, precision=PREC_F32, device=0):
    L = load()
    x = np.ascontiguousarray(x, np.float32); w = np.ascontiguousarray(w_hwio, np.float32); b = np.ascontiguousarray(bias, np.float32)
    n, h, wd, cin = x.shape
    ks, _, _, cout = w.shape
    ho = (h + 2 * (ks // 2) - ks) // stride + 1
    wo = (wd + 2 * (ks // 2) - ks) // stride + 1
    y = np.empty((n, ho, wo, cout), np.float32)
    r1 = None if r1 is None else np.ascontiguousarray(r1, np.float32)
    r2 = None if r2 is None else np.ascontiguousarray(r2, np.float32)
    rc = L.eagle_op_conv2d(device, precision, _fp(x), n, h, wd, cin, _fp(w), _fp(b), cout, ks, stride, pre, _fp(r1), _fp(r2), post, _fp(y))
    if rc:
        raise EagleError(f"eagle_op_conv2d failed ({rc}): {L.eagle_last_error(None).decode()}")
    return y


def op_bottleneck(x, w1, b1, w2, b2, w3, b3, res=None, reps=0, device=0, wd=None, bd=None):
    """One fused Bottleneck launch of the split family (include/eagle.h eagle_op_bottleneck; csrc/bneck.hip).  Returns y, or (y, ms per launch) when reps > 0."""
    L = load()
    x = np.ascontiguousarray(x, np.float32)
    n, h, w, cin = x.shape
    arrs = [np.ascontiguousarray(a, np.float32) for a in (w1, b1, w2, b2, w3, b3)]
    assert arrs[0].shape == (1, 1, cin, 64) and arrs[2].shape == (3, 3, 64, 64) and arrs[4].shape == (1, 1, 64, 256), "HWIO weights of a 64-wide Bottleneck"
    res = None if res is None else np.ascontiguousarray(res, np.float32)
    y = np.empty((n, h, w, 256), np.float32)
    ms = C.c_float(0)
    wd = None if wd is None else np.ascontiguousarray(wd, np.float32)      # the 1x1 downsample branch [1][1][64][256] computed inside the launch (block 0)
    bd = None if bd is None else np.ascontiguousarray(bd, np.float32)
    rc = L.eagle_op_bottleneck(device, _fp(x), n, h, w, cin, *[_fp(a) for a in arrs], _fp(res), _fp(y), int(reps), C.byref(ms), _fp(wd), _fp(bd))
    if rc:
        raise EagleError(f"eagle_op_bottleneck failed ({rc}): {L.eagle_last_error(None).decode()}")
    return (y, ms.value) if reps > 0 else y


def op_fuse_sum(base, ups, relu=True, precision=PREC_F32, device=0):
    L = load()
    base = np.ascontiguousarray(base, np.float32)
    n, H, W, c = base.shape
    ups = [np.ascontiguousarray(u, np.float32) for u in ups]
    arr = (C.POINTER(C.c_float) * max(len(ups), 1))(*[_fp(u) for u in ups])
    uh = (C.c_int * max(len(ups), 1))(*[u.shape[1] for u in ups])
    uw = (C.c_int * max(len(ups), 1))(*[u.shape[2] for u in ups])
    y = np.empty_like(base)
    rc = L.eagle_op_fuse_sum(device, precision, _fp(base), n, H, W, c, len(ups), arr, uh, uw, int(relu), _fp(y))
    if rc:
        raise EagleError(f"eagle_op_fuse_sum failed ({rc}): {L.eagle_last_error(None).decode()}")
    return y


def op_preprocess(frames, det_imgsz=640, precision=PREC_F32, device=0, letterbox=0):
    L = load()
    frames = np.ascontiguousarray(frames, np.uint8)
    n, h, w, _ = frames.shape
    hw = (C.c_int * 2)()
    rc = L.eagle_op_preprocess_lb(device, precision, frames.ctypes.data_as(C.POINTER(C.c_uint8)), n, h, w, det_imgsz, int(letterbox), None, None, hw)
    if rc:
        raise EagleError(f"eagle_op_preprocess failed ({rc}): {L.eagle_last_error(None).decode()}")
    kp = np.empty((n, 540, 960, 3), np.float32)
    det = np.empty((n, hw[0], hw[1], 3), np.float32)
    rc = L.eagle_op_preprocess_lb(device, precision, frames.ctypes.data_as(C.POINTER(C.c_uint8)), n, h, w, det_imgsz, int(letterbox), _fp(kp), _fp(det), hw)
    if rc:
        raise EagleError(f"eagle_op_preprocess failed ({rc}): {L.eagle_last_error(None).decode()}")
    return kp, det


def op_find_homography(img_pts, world_pts, thresh=5.0, max_iters=2000, lm_iters=10, device=0):
    L = load()
    a = np.ascontiguousarray(img_pts, np.float32).reshape(-1, 2)
    b = np.ascontiguousarray(world_pts, np.float32).reshape(-1, 2)
    n = len(a)
    H = np.zeros(9, np.float64); mask = np.zeros(max(n, 1), np.uint8); ok = C.c_int(0)
    rc = L.eagle_op_find_homography(device, _fp(a), _fp(b), n, thresh, max_iters, lm_iters,
                                    H.ctypes.data_as(C.POINTER(C.c_double)), mask.ctypes.data_as(C.POINTER(C.c_uint8)), C.byref(ok))
    if rc:
        raise EagleError(f"eagle_op_find_homography failed ({rc}): {L.eagle_last_error(None).decode()}")
    return (H.reshape(3, 3), mask[:n]) if ok.value else (None, None)

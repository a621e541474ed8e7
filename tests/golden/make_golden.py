"""Generates the committed golden fixtures by RUNNING THE REFERENCE in the build container (it cannot travel to
the GPU box).  Usage:  PYTHONPATH=/root/repo python tests/golden/make_golden.py

1. pitch_tables.json   — dumped from the reference's eagle/utils/pitch.py (pins SURVEY §8 row a14).
2. hrnet_golden.npz    — the reference's own ``KeypointModel(57)`` (eagle/models/keypoint_hrnet.py, imported by file
   path) loaded with this repo's seeded synthetic state-dict (strict=True), run on seeded inputs; stores
   ``get_keypoints`` tuples and strided logits (pins rows a2, a3).
3. loop_golden.json    — the reference's own ``CoordinateModel.get_coordinates`` loop body
   (eagle/models/coordinate_model.py:277-415) and ``detect_objects`` / ``detect_keypoints`` executed with the four
   missing third-party packages stubbed: the *control logic, integer rules and record layout are the reference's*,
   while cv2.findHomography / perspectiveTransform / fitLine are served by this repo's restatements
   (oracle/eo_prims.c) and the two networks by canned outputs.  Pins rows a4, a5 (minus fitLine), a9, a10 (point
   selection + inlier filtering), a11, a12, a13 of the host-logic oracle.
4. flow_golden.json    — the reference's own loop with keypoint_interval > 1 (optical-flow propagation, first-frame forward search,
   on-demand detection, calibration; cm.py:188-331, 419-478, 520-555) on the synthetic clips of tests/flow_cases.py, with
   cv2.cvtColor / calcOpticalFlowPyrLK served by oracle/eo_flow.c.  Pins the control logic of SURVEY §8f row 2.
Only data (inputs + expected outputs) is written; no reference source text is stored."""
import importlib.util
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = "/root/reference"

from eagle_amd import synth, weights  # noqa: E402
from oracle import prims as P  # noqa: E402


def load_by_path(name, path):
    spec = importlib.util.spec_from_file_location(name, path)
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


# ------------------------------------------------------------------------------------------------------------ 1
def dump_pitch():
    pitch = load_by_path("ref_pitch", f"{REF}/eagle/utils/pitch.py")
    out = {"INTERSECTION_TO_PITCH_POINTS": {str(k): v for k, v in pitch.INTERSECTION_TO_PITCH_POINTS.items()},
           "NOT_ON_PLANE": list(pitch.NOT_ON_PLANE),
           "GROUND_TRUTH_POINTS": [[k, list(v)] for k, v in pitch.GROUND_TRUTH_POINTS.items()]}
    json.dump(out, open(f"{HERE}/pitch_tables.json", "w"), indent=0)


# ------------------------------------------------------------------------------------------------------------ 2
def dump_hrnet():
    kh = load_by_path("ref_kh", f"{REF}/eagle/models/keypoint_hrnet.py")
    sd = weights.make_hrnet_state_dict(0)
    model = kh.KeypointModel(57).eval()
    model.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in sd.items()}, strict=True)
    out = {}
    with torch.no_grad():
        # small seeded input (odd sizes: exercises 25->13->7->4 style down/up-sampling)
        x = np.random.default_rng(11).standard_normal((2, 100, 148, 3)).astype(np.float32)
        xt = torch.from_numpy(x).permute(0, 3, 1, 2).contiguous()
        out["small_logits"] = model.forward_unnormalized(xt).permute(0, 2, 3, 1).numpy()
        kps = model.get_keypoints(xt)
        out["small_kp"] = np.array([[list(t) for t in f] + [[-1, 0, 0, 0]] * (57 - len(f)) for f in kps], np.float64)
        # one full-resolution frame, preprocessed by this repo's preprocess restatement
        from oracle import host
        xf = host.preprocess_keypoints(synth.frame(0, 0))
        xft = torch.from_numpy(xf).permute(0, 3, 1, 2).contiguous()
        lg = model.forward_unnormalized(xft).permute(0, 2, 3, 1).numpy()
        out["full_logits_strided"] = lg[:, ::8, ::8].copy()
        kpf = model.get_keypoints(xft)[0]
        out["full_kp"] = np.array([list(t) for t in kpf] + [[-1, 0, 0, 0]] * (57 - len(kpf)), np.float64)
    np.savez_compressed(f"{HERE}/hrnet_golden.npz", **out)
    print("hrnet golden:", {k: v.shape for k, v in out.items()})


# ------------------------------------------------------------------------------------------------------------ 3
def install_stubs():
    cv2 = types.ModuleType("cv2")
    cv2.COLOR_BGR2RGB, cv2.COLOR_BGR2GRAY, cv2.COLOR_BGR2HSV = 4, 6, 40
    cv2.RANSAC, cv2.RHO, cv2.LMEDS, cv2.DIST_L2 = 8, 16, 4, 2
    cv2.TERM_CRITERIA_EPS, cv2.TERM_CRITERIA_COUNT = 2, 1
    # BGR2GRAY / BGR2HSV are served by the oracle's restatements (eo_flow.c); BGR2RGB leaves the frame alone so that the
    # canned-network stubs can read the frame index from pixel [0,0,0]
    cv2.cvtColor = lambda img, code: P.bgr2gray(img) if code == 6 else (P.bgr2hsv(img) if code == 40 else img)
    cv2.calcOpticalFlowPyrLK = lambda prev, cur, pts, nxt, winSize=(15, 15), maxLevel=2, criteria=(3, 10, 0.03): (
        *P.calc_optical_flow_pyr_lk(prev, cur, pts, maxLevel, criteria[1], criteria[2]), None)

    def find_h(src, dst, method, thr=None):
        # cm.py:354-357 tries cv2.RANSAC (8), cv2.RHO (16), cv2.LMEDS (4) in turn: all three are served by the oracle's restatements
        # (a threshold of None is findHomography's default 3.0; LMEDS ignores it)
        if method == 8:
            H, mask = P.find_homography(src, dst, 8, 5.0 if thr is None else thr)
        elif method == 16:
            H, mask = P.find_homography(src, dst, 16, 3.0 if thr is None else thr)
        elif method == 4:
            H, mask = P.find_homography(src, dst, 4)
        else:
            return None, None
        return (H, mask.reshape(-1, 1)) if H is not None else (None, None)

    cv2.findHomography = find_h
    cv2.perspectiveTransform = lambda pts, H: P.perspective_transform(np.asarray(pts, np.float32).reshape(-1, 2), H).reshape(np.asarray(pts).shape)
    cv2.fitLine = lambda pts, *a: P.fit_line_l2(np.asarray(pts, np.float32).reshape(-1, 2)).reshape(4, 1)
    sys.modules["cv2"] = cv2
    ul = types.ModuleType("ultralytics"); ul.YOLO = object; sys.modules["ultralytics"] = ul
    bm = types.ModuleType("boxmot"); bm.BotSort = object; sys.modules["boxmot"] = bm
    A = types.ModuleType("albumentations")
    A.Compose = A.Resize = A.Normalize = lambda *a, **k: None
    Ap = types.ModuleType("albumentations.pytorch"); Ap.ToTensorV2 = lambda *a, **k: None
    sys.modules["albumentations"] = A; sys.modules["albumentations.pytorch"] = Ap


class _Boxes:
    def __init__(self, d):
        d = np.asarray(d, np.float32).reshape(-1, 6)
        self.xyxy, self.conf, self.cls = torch.from_numpy(d[:, :4].copy()), torch.from_numpy(d[:, 4].copy()), torch.from_numpy(d[:, 5].copy())


def make_cases():
    """(keypoint tuples per frame, detections per frame).  Geometry comes from the synthetic camera so that the
    homography is meaningful; some cases add collisions, off-plane labels, outliers, <4 plane points, no detections."""
    rng = np.random.default_rng(5)
    cases = []
    for ci in range(8):
        vis = synth.visible_landmarks(ci, 13 * ci)
        kp = []
        for idx, (x, y) in sorted(vis.items()):
            if rng.random() < 0.75:
                xn = min(239, int(round(x / 1280 * 239))) / 239
                yn = min(134, int(round(y / 720 * 134))) / 134
                kp.append((idx, xn, yn, float(np.float32(rng.uniform(0.25, 0.99)))))
        if ci == 1:   # two labels on one heat-map pixel with different scores, a third with equal score
            kp.append((30, kp[0][1], kp[0][2], float(np.float32(0.995))))
            kp.append((31, kp[1][1], kp[1][2], kp[1][3]))
        if ci == 2:   # off-plane goal-post tops are labelled but never used for H
            kp += [(0, 0.1, 0.2, 0.9), (1, 0.12, 0.2, 0.9), (24, 0.8, 0.2, 0.9)]
        if ci == 3:   # gross outliers
            kp = [(i, float(rng.uniform(0, 1)), float(rng.uniform(0, 1)), s) if k % 4 == 0 else (i, x, y, s) for k, (i, x, y, s) in enumerate(kp)]
        if ci == 4:   # fewer than 4 plane points -> no homography
            kp = kp[:2] + [(0, 0.3, 0.3, 0.8), (25, 0.6, 0.3, 0.8)]
        if ci == 5:   # low scores: some below keypoint_conf, some below 0.01
            kp = [(i, x, y, s * (0.02 if k % 3 == 0 else 1.0)) for k, (i, x, y, s) in enumerate(kp)]
        kp.sort(key=lambda t: t[0])
        nd = [12, 40, 0, 25, 9, 300, 3, 60][ci]
        d = np.zeros((nd, 6), np.float32)
        d[:, 0] = rng.uniform(-5, 1250, nd); d[:, 1] = rng.uniform(-5, 690, nd)
        d[:, 2] = d[:, 0] + rng.uniform(4, 90, nd); d[:, 3] = d[:, 1] + rng.uniform(8, 160, nd)
        d[:, [0, 2]] = d[:, [0, 2]].clip(0, 1280); d[:, [1, 3]] = d[:, [1, 3]].clip(0, 720)
        d[:, 4] = np.sort(rng.uniform(0.15, 0.97, nd))[::-1]
        d[:, 5] = rng.choice([0, 0, 0, 0, 1, 2, 3, 4], nd)
        if ci == 6:
            d[:, 5] = [2, 2, 2]; d[1, 4] = 0.2          # ball enumerate index with a gap
        cases.append((kp, d))
    return cases


def dump_loop():
    install_stubs()
    pkg = types.ModuleType("eagle"); pkg.__path__ = [f"{REF}/eagle"]; sys.modules["eagle"] = pkg
    sub = types.ModuleType("eagle.models"); sub.__path__ = [f"{REF}/eagle/models"]; sys.modules["eagle.models"] = sub
    ut = types.ModuleType("eagle.utils"); ut.__path__ = [f"{REF}/eagle/utils"]; sys.modules["eagle.utils"] = ut
    sys.modules["eagle.utils.pitch"] = load_by_path("eagle.utils.pitch", f"{REF}/eagle/utils/pitch.py")
    sys.modules["eagle.models.keypoint_hrnet"] = load_by_path("eagle.models.keypoint_hrnet", f"{REF}/eagle/models/keypoint_hrnet.py")
    cm = load_by_path("eagle.models.coordinate_model", f"{REF}/eagle/models/coordinate_model.py")
    sys.modules["eagle.models.coordinate_model"] = cm
    out = []
    for kp, dets in make_cases():
        m = object.__new__(cm.CoordinateModel)
        m.device = "cpu"
        m.class_names = {0: "Player", 1: "Goalkeeper", 2: "Ball", 3: "Referee", 4: "Staff members"}
        m.keypoint_conf, m.detector_conf = 0.3, 0.35
        m.lk_params = {}
        m.transforms = lambda image: {"image": torch.zeros(3, 4, 4)}
        km = types.SimpleNamespace()
        km.unnormalized_model = [None, types.SimpleNamespace(weight=types.SimpleNamespace(data=torch.zeros(1)))]
        km.get_keypoints = lambda x, kp=kp: [list(kp) for _ in range(x.shape[0])]
        m.keypoint_model = km
        m.detector_model = lambda frame, verbose=False, conf=0.15, dets=dets: [types.SimpleNamespace(boxes=_Boxes(dets))]
        m.tracker = types.SimpleNamespace(update=lambda d, f: np.zeros((0, 8)))
        frames = [np.zeros((720, 1280, 3), np.uint8)]
        res = m.get_coordinates(frames, fps=1, num_homography=1, num_keypoint_detection=1, verbose=False)
        objs = m.detect_objects(frames[0])
        kps_only = m.detect_keypoints(frames[0])
        out.append({"kp": [list(t) for t in kp], "dets": dets.tolist(),
                    "record": json.loads(json.dumps(res[0], default=float)),
                    "detect_objects": json.loads(json.dumps(objs, default=lambda o: o.tolist() if isinstance(o, np.ndarray) else float(o))),
                    "detect_keypoints": json.loads(json.dumps(kps_only, default=float))})
    json.dump(out, open(f"{HERE}/loop_golden.json", "w"))
    print("loop golden:", len(out), "cases;", [(len(c["record"]["Keypoints"]), c["record"]["Boundaries"][0] is not None) for c in out])


def dump_cadence():
    """The reference loop exactly as main.py:27 calls it at --fps 5: get_coordinates(frames, 5, num_homography=1,
    num_keypoint_detection=3) -> keypoint_interval 1, homography_interval 5 (H carried between scheduled frames, retry flag
    cm.py:350-367).  Frames carry their index in pixel [0,0,0] so the stubs can serve per-frame canned outputs."""
    cm = sys.modules["eagle.models.coordinate_model"]
    cases = make_cases()
    order = [4, 0, 1, 2, 3, 5, 6, 7, 0, 1, 2, 3]          # frame 0 = the "< 4 plane points" case -> retry on frame 1
    kps = [cases[k][0] for k in order]
    dets = [cases[k][1] for k in order]
    frames = []
    for i in range(len(order)):
        f = np.zeros((720, 1280, 3), np.uint8); f[0, 0, 0] = i
        frames.append(f)
    m = object.__new__(cm.CoordinateModel)
    m.device = "cpu"
    m.class_names = {0: "Player", 1: "Goalkeeper", 2: "Ball", 3: "Referee", 4: "Staff members"}
    m.keypoint_conf, m.detector_conf = 0.3, 0.35
    m.lk_params = {}
    m.transforms = lambda image: {"image": torch.full((3, 4, 4), float(image[0, 0, 0]))}
    km = types.SimpleNamespace()
    km.unnormalized_model = [None, types.SimpleNamespace(weight=types.SimpleNamespace(data=torch.zeros(1)))]
    km.get_keypoints = lambda x: [list(kps[int(x[b, 0, 0, 0])]) for b in range(x.shape[0])]
    m.keypoint_model = km
    m.detector_model = lambda frame, verbose=False, conf=0.15: [types.SimpleNamespace(boxes=_Boxes(dets[int(frame[0, 0, 0])]))]
    m.tracker = types.SimpleNamespace(update=lambda d, f: np.zeros((0, 8)))
    res = m.get_coordinates(frames, 5, num_homography=1, num_keypoint_detection=3, verbose=False)
    out = {"order": order, "fps": 5, "num_homography": 1, "num_keypoint_detection": 3,
           "kp": [[list(t) for t in k] for k in kps], "dets": [d.tolist() for d in dets],
           "records": json.loads(json.dumps(res, default=float))}
    json.dump(out, open(f"{HERE}/cadence_golden.json", "w"))
    print("cadence golden:", [(i, r["Boundaries"][0] is not None, len(r["Keypoints"])) for i, r in sorted(res.items())])


def _jsonable(o):
    if isinstance(o, np.ndarray):
        return o.tolist()
    if isinstance(o, np.integer):
        return int(o)
    return float(o)


def dump_flow():
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import flow_cases
    cm = sys.modules["eagle.models.coordinate_model"]
    out = {}
    for name, (fps, nh, nk, spec, calib) in flow_cases.CLIPS.items():
        frames = flow_cases.frames_of(name)
        kps, dets = flow_cases.canned(name)
        m = object.__new__(cm.CoordinateModel)
        m.device = "cpu"
        m.class_names = {0: "Player", 1: "Goalkeeper", 2: "Ball", 3: "Referee", 4: "Staff members"}
        m.keypoint_conf, m.detector_conf = 0.3, 0.35
        m.lk_params = dict(winSize=(15, 15), maxLevel=2, criteria=(3, 10, 0.03))
        m.transforms = lambda image: {"image": torch.full((3, 4, 4), float(image[0, 0, 0]))}
        calls = []
        km = types.SimpleNamespace()
        km.unnormalized_model = [None, types.SimpleNamespace(weight=types.SimpleNamespace(data=torch.zeros(1)))]

        def get_keypoints(x, kps=kps, calls=calls):
            calls.extend(int(x[b, 0, 0, 0]) for b in range(x.shape[0]))
            return [list(kps[int(x[b, 0, 0, 0])]) for b in range(x.shape[0])]

        km.get_keypoints = get_keypoints
        m.keypoint_model = km
        m.detector_model = lambda frame, verbose=False, conf=0.15, dets=dets: [types.SimpleNamespace(boxes=_Boxes(dets[int(frame[0, 0, 0])]))]
        m.tracker = types.SimpleNamespace(update=lambda d, f: np.zeros((0, 8)))
        try:
            res = m.get_coordinates(frames, fps, num_homography=nh, num_keypoint_detection=nk, verbose=False, calibration=calib)
            rec = json.loads(json.dumps(res, default=_jsonable))
            err = None
        except Exception as e:                       # the reference's own exception (calibration near the image border)
            rec, err = None, type(e).__name__
        out[name] = {"records": rec, "raises": err, "detected_frames": sorted(set(calls))}
        print("flow golden:", name, err or [(i, len(r["Keypoints"]), r["Boundaries"][0] is not None) for i, r in sorted(res.items())],
              "detect on", sorted(set(calls)))
    json.dump(out, open(f"{HERE}/flow_golden.json", "w"))


def dump_ingest():
    """ingest_golden.json: the reference's own read_video (eagle/utils/io.py:5-27) over a stubbed cv2.VideoCapture that yields the
    frame index as the frame: which frames it keeps for a native / target frame-rate pair (pins the sampling of SURVEY §8f row 4)."""
    cv2 = sys.modules["cv2"]
    cv2.CAP_PROP_FPS = 5

    class Cap:
        spec = (0, 0.0)

        def __init__(self, path):
            self.n, self.fps = Cap.spec
            self.k = 0

        def get(self, prop):
            return self.fps

        def read(self):
            if self.k >= self.n:
                return False, None
            self.k += 1
            return True, self.k - 1

        def release(self):
            pass

    cv2.VideoCapture = Cap
    io = load_by_path("ref_io", f"{REF}/eagle/utils/io.py")
    out = []
    for n, native, fps in [(50, 25.0, 24), (50, 29.97, 24), (60, 50.0, 24), (61, 60.0, 24), (30, 59.94, 5), (10, 24.0, 5), (7, 30.0, 30), (9, 23.976, 24)]:
        Cap.spec = (n, native)
        try:
            frames, _ = io.read_video(f"{REF}/main.py", fps)
            out.append({"n": n, "native_fps": native, "fps": fps, "kept": frames, "raises": None})
        except Exception as e:
            out.append({"n": n, "native_fps": native, "fps": fps, "kept": None, "raises": type(e).__name__})
    json.dump(out, open(f"{HERE}/ingest_golden.json", "w"))


# ------------------------------------------------------------------------------------------------------------ 7
def dump_team():
    """Team colours: the reference's OWN Processor.get_team_mapping / detect_color (eagle/processor.py:405-503) over synthetic frames and
    ground-truth player boxes, with cv2 replaced by numpy definitions (inRange, bitwise_and, countNonZero, BGR2RGB) and the oracle's
    BGR2HSV restatement; scikit-learn's KMeans is the real one."""
    import importlib.util
    import types as _types
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import team_cases
    cv2 = _types.ModuleType("cv2")
    cv2.COLOR_BGR2RGB, cv2.COLOR_BGR2HSV = 4, 40
    cv2.cvtColor = lambda img, code: P.bgr2hsv(np.ascontiguousarray(img)) if code == 40 else np.ascontiguousarray(img[..., ::-1])

    def bitwise_and(a, b, mask=None):
        out = np.bitwise_and(a, b)
        if mask is not None:
            out = np.where((mask != 0)[..., None] if out.ndim == 3 else (mask != 0), out, 0).astype(a.dtype)
        return out
    cv2.bitwise_and = bitwise_and
    cv2.inRange = lambda img, lo, hi: (np.all((img >= lo) & (img <= hi), axis=2).astype(np.uint8) * 255)
    cv2.countNonZero = lambda m: int(np.count_nonzero(m))
    cv2.KalmanFilter = object
    saved = sys.modules.get("cv2")
    sys.modules["cv2"] = cv2
    try:
        spec = importlib.util.spec_from_file_location("ref_processor", os.path.join(REF, "eagle", "processor.py"))
        proc = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(proc)
    finally:
        if saved is not None:
            sys.modules["cv2"] = saved
    frames, coords = team_cases.make_case()
    pr = proc.Processor(coords, frames, fps=5)
    mapping = pr.get_team_mapping()
    crops = []
    for i, frame in enumerate(frames):
        for pid, it in coords[i]["Coordinates"]["Player"].items():
            x1, y1, x2, y2 = it["BBox"]
            crops.append({"frame": i, "player": int(pid), "bbox": it["BBox"], "colors": [[c, int(n)] for c, n in pr.detect_color(frame[y1:y2, x1:x2])]})
    json.dump({"team_mapping": {str(k): int(v) for k, v in mapping.items()}, "crops": crops}, open(f"{HERE}/team_golden.json", "w"))
    print("team golden:", len(crops), "crops,", mapping)


if __name__ == "__main__":
    dump_pitch()
    dump_hrnet()
    dump_loop()
    dump_cadence()
    dump_flow()
    dump_ingest()
    dump_team()

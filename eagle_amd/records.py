"""EagleFrameResult -> the reference's per-frame record ``res[i]`` (eagle/models/coordinate_model.py:369-415,
schema docs/data.md:20-41).  Pure bookkeeping on the host: every number was computed on the GPU."""
import numpy as np

from .pitch import INTERSECTION_TO_PITCH_POINTS

CLASS_NAMES = {0: "Player", 1: "Goalkeeper", 2: "Ball", 3: "Referee", 4: "Staff members"}  # cm.py:61


def _objects(rec):
    """cm.py:598-627 ordering: Player and Goalkeeper keyed by detection index, then Ball keyed by enumerate index."""
    out = {}
    dets = rec["det"][: int(rec["n_det"])]
    for want in (0, 1, 2):
        for d in dets:
            if int(d["cls"]) != want or not d["reported"]:
                continue
            out.setdefault(CLASS_NAMES[want], {})[int(d["id"])] = d
    return out


def to_reference_dict(rec, i=0, fps=25, own_h=True):
    """One record -> {"Coordinates", "Time", "Keypoints", "Boundaries"} exactly as cm.py:415 builds it.
    own_h=False: the frame uses a carried homography (cadence mode): key-points are not inlier-filtered (cm.py:330,415)."""
    H_valid = bool(rec["H_valid"])
    coords = {}
    for cname, objs in _objects(rec).items():
        for oid, d in objs.items():
            bbox = [int(d["bx1"]) & 0xFFFF, int(d["by1"]) & 0xFFFF, int(d["bx2"]) & 0xFFFF, int(d["by2"]) & 0xFFFF]  # uint16 cast, cm.py:373
            cur = {"BBox": bbox, "Confidence": float(d["conf"])}
            if H_valid and d["in_bounds"]:
                cur["Transformed_Coordinates"] = [int(d["pitch_x"]), int(d["pitch_y"])]
            else:
                cur["Transformed_Coordinates"] = None
                cur["Image_Bottom_center"] = [int(d["foot_x"]), int(d["foot_y"])]
            coords.setdefault(cname, {})[oid] = cur
    kps = rec["kp"][: int(rec["n_kp"])]
    if H_valid and own_h:   # cm.py:359-362: inliers only, values come back from img_pts.tolist() as floats
        keypoints = {INTERSECTION_TO_PITCH_POINTS[int(k["label"])]: [float(k["x"]), float(k["y"])]
                     for k in kps if k["on_plane"] and k["inlier"]}
    else:
        # values that came out of the optical flow (cm.py:476) or were moved by the calibration (cm.py:551-553) are numpy integers in
        # the reference, and main.py's json.dump(default=float) writes those as floats: keep the type so the file is identical
        keypoints = {INTERSECTION_TO_PITCH_POINTS[int(k["label"])]: ((np.int64(k["x"]), np.int64(k["y"])) if k["pad"] else (int(k["x"]), int(k["y"])))
                     for k in kps}
    if rec["bounds_valid"]:
        b = rec["bounds"]
        bounds = [(float(b[0]), 0), (float(b[1]), 68), (float(b[2]), 68), (float(b[3]), 0)]
    else:
        bounds = [None] * 4
    return {"Coordinates": coords, "Time": f"{i // fps // 60:02d}:{i // fps % 60:02d}", "Keypoints": keypoints, "Boundaries": bounds}


def to_process_dict(rec):
    """The north_star's ``Processor.process(frame) -> {players, ball, H}`` view of one record."""
    ref = to_reference_dict(rec)
    players = {}
    for cname in ("Player", "Goalkeeper"):
        for oid, d in ref["Coordinates"].get(cname, {}).items():
            players[oid] = dict(d, Type=cname)
    return {"players": players, "ball": ref["Coordinates"].get("Ball", {}),
            "H": np.array(rec["H"], np.float64).reshape(3, 3) if rec["H_valid"] else None,
            "keypoints": ref["Keypoints"], "boundaries": ref["Boundaries"]}

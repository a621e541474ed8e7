"""ORACLE — test infrastructure only (see eo_prims.c header).

Python restatement of the non-network steps of the reference's per-frame loop body
(eagle/models/coordinate_model.py:277-415) in the stateless configuration of SURVEY §8:
keypoint_interval = homography_interval = 1, calibration off, tracker off (IDs = detection index,
coordinate_model.py:598-627).  Each function names the lines it follows.  Pure Python / numpy on purpose:
it reads like the reference so that it can be diffed against it; arithmetic that must be bit-reproducible
on the GPU (exp, fitLine, DLT) lives in eo_prims.c.
"""
import math
from collections import Counter

import numpy as np

from eagle_amd.pitch import (GROUND_TRUTH_POINTS, INTERSECTION_TO_PITCH_POINTS, NOT_ON_PLANE,
                             PITCH_POINTS_TO_INTERSECTION, PITCH_HEIGHT, PITCH_WIDTH)
from . import prims as P

CLASS_NAMES = {0: "Player", 1: "Goalkeeper", 2: "Ball", 3: "Referee", 4: "Staff members"}  # cm.py:61

# A.Normalize() constants exactly as albumentations builds them (float32): mean*255, 1/(std*255)
_MEAN255 = (np.array([0.485, 0.456, 0.406], np.float32) * np.float32(255.0)).astype(np.float32)
_INVSTD = np.reciprocal(np.array([0.229, 0.224, 0.225], np.float32) * np.float32(255.0)).astype(np.float32)


# ----------------------------------------------------------------------------------------------------------
# a1: keypoint-net preprocessing  (cm.py:489-491 with the transforms of cm.py:62-64)
# ----------------------------------------------------------------------------------------------------------
def preprocess_keypoints(frame_bgr, out_h=540, out_w=960):
    """u8 HWC BGR -> float32 [1,540,960,3] RGB, ImageNet-normalised."""
    rgb = np.ascontiguousarray(frame_bgr[:, :, ::-1])
    r = P.resize_linear_u8c3(rgb, out_h, out_w).astype(np.float32)
    return ((r - _MEAN255) * _INVSTD)[None]


# ----------------------------------------------------------------------------------------------------------
# a6: detector preprocessing (ultralytics LetterBox + BGR->RGB + /255)
# ----------------------------------------------------------------------------------------------------------
def letterbox_geometry(h, w, imgsz=640, stride=32, auto=True):
    r = min(imgsz / h, imgsz / w)
    new_w, new_h = int(round(w * r)), int(round(h * r))
    dw, dh = imgsz - new_w, imgsz - new_h
    if auto:
        dw, dh = dw % stride, dh % stride
    dw /= 2
    dh /= 2
    top, bottom = int(round(dh - 0.1)), int(round(dh + 0.1))
    left, right = int(round(dw - 0.1)), int(round(dw + 0.1))
    return dict(new_w=new_w, new_h=new_h, top=top, left=left, out_h=new_h + top + bottom, out_w=new_w + left + right)


def preprocess_detector(frame_bgr, imgsz=640, auto=True):
    """auto=True: the .pt predictor's LetterBox (pad to a multiple of 32; cm.py:56-57); auto=False: the static imgsz x imgsz input an exported ONNX detector
    runs with (the reference's CPU default, cm.py:54-55 — ultralytics' AutoBackend sets auto = pt, i.e. False for .onnx)."""
    h, w = frame_bgr.shape[:2]
    g = letterbox_geometry(h, w, imgsz, auto=auto)
    rgb = np.ascontiguousarray(frame_bgr[:, :, ::-1])
    r = P.resize_linear_u8c3(rgb, g["new_h"], g["new_w"])
    canvas = np.full((g["out_h"], g["out_w"], 3), 114, np.uint8)
    canvas[g["top"]:g["top"] + g["new_h"], g["left"]:g["left"] + g["new_w"]] = r
    return (canvas.astype(np.float32) / np.float32(255.0))[None], g


# ----------------------------------------------------------------------------------------------------------
# a8: non_max_suppression + scale_boxes (ultralytics; torchvision.ops.nms semantics)
# ----------------------------------------------------------------------------------------------------------
def nms_and_scale(rows, frame_h, frame_w, in_h, in_w, conf_thres=0.15, iou_thres=0.7, max_det=300, max_wh=7680.0):
    """rows [A, 4+nc] (cx,cy,w,h,cls...) float32 -> [K,6] float32 (x1,y1,x2,y2,conf,cls), K order = descending conf."""
    rows = np.asarray(rows, np.float32)
    cx, cy, bw, bh = rows[:, 0], rows[:, 1], rows[:, 2], rows[:, 3]
    hw, hh = bw / np.float32(2), bh / np.float32(2)
    box = np.stack([cx - hw, cy - hh, cx + hw, cy + hh], 1).astype(np.float32)
    cls = rows[:, 4:]
    conf = cls.max(1)
    j = cls.argmax(1)
    keep = conf > np.float32(conf_thres)
    idx = np.nonzero(keep)[0]
    order = idx[np.argsort(-conf[idx], kind="stable")]          # descending conf, ties by anchor index
    box, conf, j = box[order], conf[order], j[order]
    off = (j.astype(np.float32) * np.float32(max_wh))[:, None]
    b = (box + off).astype(np.float32)
    area = ((b[:, 2] - b[:, 0]) * (b[:, 3] - b[:, 1])).astype(np.float32)
    n = len(order)
    dead = np.zeros(n, bool)
    kept = []
    for i in range(n):
        if dead[i]:
            continue
        kept.append(i)
        if len(kept) >= max_det:
            break
        xx1 = np.maximum(b[i, 0], b[i + 1:, 0]); yy1 = np.maximum(b[i, 1], b[i + 1:, 1])
        xx2 = np.minimum(b[i, 2], b[i + 1:, 2]); yy2 = np.minimum(b[i, 3], b[i + 1:, 3])
        iw = np.maximum(np.float32(0), xx2 - xx1); ih = np.maximum(np.float32(0), yy2 - yy1)
        inter = (iw * ih).astype(np.float32)
        with np.errstate(divide="ignore", invalid="ignore"):
            ovr = inter / (area[i] + area[i + 1:] - inter)
        dead[i + 1:] |= ovr > np.float32(iou_thres)
    kept = np.array(kept, np.int64)
    out = np.concatenate([box[kept], conf[kept, None], j[kept, None].astype(np.float32)], 1).astype(np.float32)
    # scale_boxes
    gain = min(in_h / frame_h, in_w / frame_w)
    pad_x = round((in_w - frame_w * gain) / 2 - 0.1)
    pad_y = round((in_h - frame_h * gain) / 2 - 0.1)
    out[:, [0, 2]] -= np.float32(pad_x)
    out[:, [1, 3]] -= np.float32(pad_y)
    out[:, :4] /= np.float32(gain)
    out[:, [0, 2]] = out[:, [0, 2]].clip(0, frame_w)
    out[:, [1, 3]] = out[:, [1, 3]].clip(0, frame_h)
    return out


# ----------------------------------------------------------------------------------------------------------
# a9: detections -> object dict, tracker-less ID scheme (cm.py:598-627)
# ----------------------------------------------------------------------------------------------------------
def objects_from_detections(dets, frame_h, frame_w, detector_conf=0.35):
    coords, conf, labels = dets[:, :4], dets[:, 4], dets[:, 5].astype(int)
    res = {"Player": {}, "Goalkeeper": {}}
    for det_i in range(coords.shape[0]):
        label_str = CLASS_NAMES.get(int(labels[det_i]))
        if label_str not in res:
            continue
        x1, y1, x2, y2 = coords[det_i].astype(int)
        x1 = int(np.clip(x1, 0, frame_w - 1)); y1 = int(np.clip(y1, 0, frame_h - 1))
        x2 = int(np.clip(x2, 0, frame_w - 1)); y2 = int(np.clip(y2, 0, frame_h - 1))
        if float(conf[det_i]) < detector_conf:
            continue
        res[label_str][det_i] = {"BBox": [x1, y1, x2, y2], "Confidence": float(conf[det_i]),
                                 "Bottom_center": [int((x1 + x2) / 2), y2]}
    if 2 in labels:
        for i, idx in enumerate(np.where(labels == 2)[0]):
            box = coords[idx].astype(int)
            res.setdefault("Ball", {})
            if float(conf[idx]) < detector_conf:
                continue
            res["Ball"][i] = {"BBox": [int(v) for v in box], "Confidence": float(conf[idx]),
                              "Bottom_center": [int((box[0] + box[2]) / 2), int(box[3])]}
    return res


# ----------------------------------------------------------------------------------------------------------
# a3 + a4: heat-map decode, threshold, label, dedup (kh.py:583-594, cm.py:500-518)
# ----------------------------------------------------------------------------------------------------------
def decode_heatmaps(idx, score, hm_h, hm_w):
    """first-max flat indices + scores per channel -> list of (i, x_n, y_n, score) with score > 0.01"""
    out = []
    for i in range(len(idx)):
        y, x = divmod(int(idx[i]), hm_w)
        s = float(score[i])
        if s > 0.01:
            out.append((i, x / max(1, hm_w - 1), y / max(1, hm_h - 1), s))
    return out


def keypoints_from_decoded(decoded, frame_h, frame_w, keypoint_conf=0.3):
    tmp = {}
    for i, x, y, score in decoded:
        if score < keypoint_conf:
            continue
        tmp[INTERSECTION_TO_PITCH_POINTS[i]] = (int(x * frame_w), int(y * frame_h), score, i)
    vals = list(tmp.values())
    counts = Counter([v[:2] for v in vals])
    coords_to_label = {}
    for k, v in tmp.items():
        if counts[v[:2]] == 1:
            coords_to_label[v[:2]] = k
        elif v[2] == max(x[2] for x in vals if x[:2] == v[:2]):
            coords_to_label[v[:2]] = k
    return {coords_to_label[c]: c for c in coords_to_label}


# ----------------------------------------------------------------------------------------------------------
# a5: keypoint synthesis by line intersection (cm.py:76-186)
# ----------------------------------------------------------------------------------------------------------
def _pitch_groups():
    coord_to_label, x_groups, y_groups = {}, {}, {}
    for label, (x, y, z) in GROUND_TRUTH_POINTS.items():
        if z != 0.0:
            continue
        xr, yr = round(float(x), 2), round(float(y), 2)
        coord_to_label.setdefault((xr, yr), label)
        x_groups.setdefault(xr, []).append(label)
        y_groups.setdefault(yr, []).append(label)
    # canonical within-group order: ascending heat-map index (the reference iterates a Python set)
    for g in list(x_groups.values()) + list(y_groups.values()):
        g.sort(key=lambda l: PITCH_POINTS_TO_INTERSECTION[l])
    return coord_to_label, x_groups, y_groups


_GROUPS = _pitch_groups()


def fit_line(points):
    if points is None or len(points) < 2:
        return None
    vx, vy, x0, y0 = (float(v) for v in P.fit_line_l2(np.asarray(points, np.float32)))
    if abs(vx) + abs(vy) < 1e-6:
        return None
    return vx, vy, x0, y0


def intersect_lines(line1, line2):
    if line1 is None or line2 is None:
        return None
    vx1, vy1, x01, y01 = line1
    vx2, vy2, x02, y02 = line2
    det = vx1 * (-vy2) - vy1 * (-vx2)
    if abs(det) < 1e-8:
        return None
    # 2x2 LU with partial pivoting, as LAPACK dgesv (np.linalg.solve) does it
    a00, a01, a10, a11 = vx1, -vx2, vy1, -vy2
    b0, b1 = x02 - x01, y02 - y01
    if abs(a10) > abs(a00):
        a00, a01, a10, a11, b0, b1 = a10, a11, a00, a01, b1, b0
    if a00 == 0.0:
        return None
    l = a10 * (1.0 / a00)
    u11 = a11 - l * a01
    if u11 == 0.0:
        return None
    t1 = (b1 - l * b0) / u11
    t = (b0 - a01 * t1) / a00
    return float(x01 + t * vx1), float(y01 + t * vy1)


def synthesize_keypoints(keypoints, min_points_per_line=2, max_new_points=30):
    coord_to_label, x_groups, y_groups = _GROUPS
    detected = {k: v for k, v in keypoints.items() if PITCH_POINTS_TO_INTERSECTION.get(k, -1) not in NOT_ON_PLANE}
    lines_y, lines_x = {}, {}
    for groups, lines in ((y_groups, lines_y), (x_groups, lines_x)):
        for val, labels in groups.items():
            pts = [detected[l] for l in labels if l in detected]
            if len(pts) >= min_points_per_line:
                line = fit_line(np.array(pts, dtype=np.float32))
                if line is not None:
                    lines[val] = line
    added = {}
    for y_val, ly in lines_y.items():
        for x_val, lx in lines_x.items():
            label = coord_to_label.get((round(float(x_val), 2), round(float(y_val), 2)))
            if not label or label in keypoints:
                continue
            pt = intersect_lines(ly, lx)
            if pt is None:
                continue
            added[label] = (int(round(pt[0])), int(round(pt[1])))
            if len(added) >= max_new_points:
                break
        if len(added) >= max_new_points:
            break
    return {**keypoints, **added} if added else keypoints


# ----------------------------------------------------------------------------------------------------------
# a10: homography (cm.py:333-367)
# ----------------------------------------------------------------------------------------------------------
def select_plane_points(keypoints):
    img_pts, world_pts, used = [], [], []
    for label, (xi, yi) in keypoints.items():
        if PITCH_POINTS_TO_INTERSECTION.get(label, -1) in NOT_ON_PLANE:
            continue
        wx, wy, wz = GROUND_TRUTH_POINTS[label]
        if wz != 0.0:
            continue
        img_pts.append([xi, yi]); world_pts.append([wx, wy]); used.append(label)
    return np.array(img_pts, np.float32).reshape(-1, 2), np.array(world_pts, np.float32).reshape(-1, 2), used


def solve_homography(keypoints):
    """-> (H | None, keypoints after inlier filtering)"""
    img_pts, world_pts, used = select_plane_points(keypoints)
    if len(img_pts) < 4:
        return None, keypoints
    H, mask = P.find_homography_ransac(img_pts, world_pts, 5.0)          # cm.py:354-357: cv2.RANSAC ...
    if H is None:
        H, mask = P.find_homography(img_pts, world_pts, 16, 3.0)         # ... cv2.RHO (threshold None = the default 3.0; restated, parity unpinned) ...
    if H is None:
        H, mask = P.find_homography(img_pts, world_pts, 4)               # ... cv2.LMEDS
    if H is None:
        return None, keypoints
    kept = {k: v for k, v, m in zip(used, img_pts.tolist(), mask.flatten()) if m}
    return H, kept


# ----------------------------------------------------------------------------------------------------------
# a11 + a12: projection of foot points and visible-pitch boundaries (cm.py:369-414, find_x_at_y cm.py:32-44)
# ----------------------------------------------------------------------------------------------------------
def find_x_at_y(pt1, pt2, y_target):
    x1, y1 = pt1
    x2, y2 = pt2
    m = (y2 - y1) / (x2 - x1)
    c = y1 - m * x1
    return (y_target - c) / m


def project_objects(objects, H):
    indiv = {}
    for class_name, class_dict in objects.items():
        for obj_id, obj in class_dict.items():
            bc = obj["Bottom_center"]
            bbox = np.array(obj["BBox"], dtype=np.uint16).tolist()
            if H is None:
                cur = {"BBox": bbox, "Confidence": obj["Confidence"], "Transformed_Coordinates": None, "Image_Bottom_center": bc}
            else:
                tf = P.perspective_transform(np.array([bc], np.float32), H)
                t = tf.astype(int)
                tx, ty = int(t[0, 0]), int(t[0, 1])
                if tx < 0 or tx > PITCH_WIDTH or ty < 0 or ty > PITCH_HEIGHT:
                    cur = {"BBox": bbox, "Confidence": obj["Confidence"], "Transformed_Coordinates": None, "Image_Bottom_center": bc}
                else:
                    cur = {"BBox": bbox, "Confidence": obj["Confidence"], "Transformed_Coordinates": [tx, ty],
                           "_pitch_float": [float(tf[0, 0]), float(tf[0, 1])]}
            indiv.setdefault(class_name, {})[int(obj_id)] = cur
    return indiv


def boundaries(H, frame_h, frame_w):
    if H is None:
        return [None] * 4
    c = P.perspective_transform(np.array([[0, 0], [frame_w, 0], [0, frame_h], [frame_w, frame_h]], np.float32), H).astype(int)
    tl, tr, bl, br = (c[i].tolist() for i in range(4))
    try:
        tl = (find_x_at_y(tl, bl, PITCH_HEIGHT), PITCH_HEIGHT)
        tr = (find_x_at_y(tr, br, PITCH_HEIGHT), PITCH_HEIGHT)
        bl = (find_x_at_y(bl, tl, 0), 0)
        br = (find_x_at_y(br, tr, 0), 0)
        return [bl, tl, tr, br]
    except Exception:
        return [None] * 4

"""ORACLE — test infrastructure only (see eo_prims.c / eo_flow.c headers).

The stateful key-point cadence of the reference loop: optical-flow propagation of key-points between
HRNet detections (eagle/models/coordinate_model.py:188-331, 419-478) and the brightness calibration (cm.py:520-555),
written to read like cm.py.  cv2 is served by oracle/eo_flow.c (parity unpinned, see its header); the control logic is
pinned to the reference's own loop by tests/golden/flow_golden.json (tests/golden/make_golden.py::dump_flow)."""
import numpy as np

from . import host
from . import prims as P


def calculate_optical_flow(frame, prev_gray, prev_keypoints, curr_gray):
    """cm.py:419-478, including its label bookkeeping: after the status filter, row j of the surviving points is paired
    with the j-th key of the UNFILTERED dict (cm.py:446), so a lost point shifts the labels of the rows after it."""
    if prev_gray is None or curr_gray is None or prev_keypoints is None or len(prev_keypoints) == 0:
        return {}
    prev_points = np.array(list(prev_keypoints.values()), dtype=np.float32)
    if prev_points.ndim != 2 or prev_points.shape[0] == 0 or prev_points.shape[1] != 2:
        return {}
    new_points, status = P.calc_optical_flow_pyr_lk(prev_gray, curr_gray, prev_points)
    new_points = new_points[status[:, 0] == 1]
    prev_points = prev_points[status[:, 0] == 1]
    filtered = {}
    move_amounts = np.linalg.norm(new_points - prev_points, axis=1)
    with np.errstate(all="ignore"):
        mean_move = np.mean(move_amounts)
        std_move = np.std(move_amounts) + 1e-6
    keys = list(prev_keypoints.keys())
    h, w = frame.shape[:2]

    def hue_at(pt):
        x, y = pt.astype(int)
        x = np.clip(x, 0, w - 1); y = np.clip(y, 0, h - 1)
        grid = frame[max(0, y - 1):min(h, y + 2), max(0, x - 1):min(w, x + 2)]
        return np.mean(P.bgr2hsv(grid)[:, :, 0])

    for j, (point, new_point) in enumerate(zip(prev_points, new_points)):
        key = keys[j]
        if (move_amounts[j] - mean_move) / std_move > 2:
            continue
        if abs(hue_at(new_point) - hue_at(point)) > 25:
            continue
        filtered[key] = tuple(new_point.astype(int))
    return filtered


def calibrate_keypoints(frame, keypoints):
    """cm.py:520-555.  (The reference indexes grid_hsv[3, 3] of a grid clipped at the image border: within 3 pixels of the
    left/top edge that is the wrong pixel, and a grid smaller than 4x4 raises IndexError there — reproduced.)"""
    OFFSET, THR = 3, 150
    out = {}
    h, w = frame.shape[:2]
    for key, (x, y) in keypoints.items():
        if not (0 <= x < w and 0 <= y < h):
            out[key] = (x, y)
            continue
        x, y = int(x), int(y)
        if int(frame[y, x].max()) >= THR:                       # V of HSV = max(B, G, R)
            out[key] = (x, y)
            continue
        grid = frame[max(0, y - OFFSET):min(h, y + OFFSET), max(0, x - OFFSET):min(w, x + OFFSET)]
        bright = grid.max(axis=2)
        _ = bright[OFFSET, OFFSET]                              # IndexError exactly where the reference raises it
        by, bx = np.unravel_index(np.argmax(bright), bright.shape)
        out[key] = (int(np.clip(x + bx - OFFSET, 0, w - 1)), int(np.clip(y + by - OFFSET, 0, h - 1)))
    return out


def loop_records(frames, fps, num_homography, num_keypoint_detection, detect_keypoints, detect_objects, calibration=False):
    """The whole reference loop (cm.py:188-416) for any cadence.  detect_keypoints(i) / detect_objects(i) return what the
    reference's methods of those names return for frame i.  Returns (res, stats)."""
    homography_interval = max(1, int(fps / max(1, num_homography)))
    keypoint_interval = max(1, int(fps / max(1, num_keypoint_detection)))
    n = len(frames)
    gray = {}

    def g(i):
        if i not in gray:
            gray[i] = P.bgr2gray(frames[i])
        return gray[i]

    stats = {"detect_calls": [], "flow_calls": 0}

    def det(i):
        stats["detect_calls"].append(i)
        return detect_keypoints(i)

    prev_gray, prev_keypoints, res = None, {}, {}
    mem = {idx: det(idx) for idx in range(0, n, keypoint_interval)}       # cm.py:217-276: batched up front
    compute_homography, H = False, None
    for i in range(n):
        frame = frames[i]
        curr_gray = g(i)
        if i == 0 or i % keypoint_interval == 0:
            keypoints = mem[i] if i in mem else det(i)
            mem[i] = keypoints
            if len(keypoints) < 4:
                if i == 0:
                    j = None
                    for j in range(i + 1, n):
                        next_gray = g(j)
                        nk = mem[j] if j in mem else det(j)
                        mem[j] = nk
                        if len(nk) >= 4:
                            prev_keypoints = nk
                            break
                    if len(prev_keypoints) > 0:
                        for j in range(j - 1, i - 1, -1):
                            pg = g(j)
                            stats["flow_calls"] += 1
                            flowed = calculate_optical_flow(frames[j], pg, prev_keypoints, next_gray)
                            prev_keypoints = flowed if len(flowed) > 0 else prev_keypoints
                            mem[j] = {**prev_keypoints, **mem.get(j, {})}
                            next_gray = pg
                else:
                    stats["flow_calls"] += 1
                    keypoints = {**keypoints, **calculate_optical_flow(frame, prev_gray, prev_keypoints, curr_gray)}
        else:
            stats["flow_calls"] += 1
            flow = calculate_optical_flow(frame, prev_gray, prev_keypoints, curr_gray)
            if len(flow) < 4:
                keypoints = mem[i] if i in mem else det(i)
                mem[i] = keypoints
                keypoints = {**keypoints, **flow}
            else:
                keypoints = {**flow, **mem.get(i, {})}
        keypoints = {**keypoints, **mem.get(i, {})}
        if len(keypoints) >= 2:
            keypoints = host.synthesize_keypoints(keypoints)
        if calibration:
            keypoints = calibrate_keypoints(frame, keypoints)
        prev_keypoints = keypoints
        prev_gray = curr_gray
        objects = detect_objects(i)
        if i % homography_interval == 0 or compute_homography:
            img_pts, world_pts, used = host.select_plane_points(keypoints)
            if len(img_pts) < 4:
                compute_homography = True
            else:
                Hn, mask = P.find_homography_ransac(img_pts, world_pts, 5.0)
                if Hn is None:                                          # cm.py:354-357: RHO is not restated, LMEDS is
                    Hn, mask = P.find_homography(img_pts, world_pts, 4)
                if Hn is not None:
                    prev_keypoints = {k: v for k, v, m in zip(used, img_pts.tolist(), mask.flatten()) if m}
                    H, compute_homography = Hn, False
                else:
                    compute_homography = True
        res[i] = {"Coordinates": host.project_objects(objects, H), "Time": f"{i // fps // 60:02d}:{i // fps % 60:02d}",
                  "Keypoints": prev_keypoints, "Boundaries": host.boundaries(H, frames[i].shape[0], frames[i].shape[1])}
    return res, stats

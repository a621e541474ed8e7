"""ORACLE — test infrastructure only (see eo_prims.c header).

S(frame): one iteration of the reference loop body (eagle/models/coordinate_model.py:277-415) in the
stateless configuration.  Returns the reference's per-frame record (cm.py:415) plus the
intermediate values the parity tests compare against the HIP path."""
import numpy as np

from . import host, nets
from . import prims as P


class OracleModel:
    """Mirror of ``CoordinateModel`` (cm.py:47-74) over synthetic state-dicts; backend 'c' (exact order) or
    'torch' (fast, MKLDNN order)."""

    def __init__(self, hrnet_sd, yolo_sd, variant="n", imgsz=640, backend="c", f16=False,
                 keypoint_conf=0.3, detector_conf=0.35, letterbox="rect"):
        self.hr_sd, self.yo_sd, self.variant, self.imgsz = hrnet_sd, yolo_sd, variant, imgsz
        assert letterbox in ("rect", "square")
        self.letterbox_auto = letterbox == "rect"      # "square": LetterBox(auto=False), the exported ONNX detector's static input (cm.py:54-55)
        self.backend, self.f16 = backend, f16
        self.keypoint_conf, self.detector_conf = keypoint_conf, detector_conf
        self._hp = nets.Params(hrnet_sd, 1e-5, f16 and backend == "c")
        self._yp = nets.Params(yolo_sd, 1e-3, f16 and backend == "c")

    # cm.py:557-628
    def detect_objects(self, frame):
        h, w = frame.shape[:2]
        x, g = host.preprocess_detector(frame, self.imgsz, auto=self.letterbox_auto)
        heads = nets.yolo_heads(self.yo_sd, x, self.variant, self.backend, self.f16, self._yp)
        rows = nets.yolo_decode(heads)
        dets = host.nms_and_scale(rows, h, w, g["out_h"], g["out_w"], conf_thres=min(self.detector_conf, 0.15))
        return host.objects_from_detections(dets, h, w, self.detector_conf), dets, rows

    # cm.py:480-518
    def detect_keypoints(self, frame):
        h, w = frame.shape[:2]
        x = host.preprocess_keypoints(frame)
        logits = nets.hrnet_logits(self.hr_sd, x, self.backend, self.f16, self._hp)
        idx, score = P.heatmap_argmax(logits[0], 57)
        decoded = host.decode_heatmaps(idx, score, logits.shape[1], logits.shape[2])
        return host.keypoints_from_decoded(decoded, h, w, self.keypoint_conf), idx, score, logits

    def step(self, frame, i=0, fps=25):
        h, w = frame.shape[:2]
        kps, idx, score, logits = self.detect_keypoints(frame)
        kp_detected = dict(kps)
        if len(kps) >= 2:
            kps = host.synthesize_keypoints(kps)
        kp_synth = dict(kps)
        objects, dets, rows = self.detect_objects(frame)
        H, kps = host.solve_homography(kps)
        indiv = host.project_objects(objects, H)
        bounds = host.boundaries(H, h, w)
        rec = {"Coordinates": indiv, "Time": f"{i // fps // 60:02d}:{i // fps % 60:02d}", "Keypoints": kps, "Boundaries": bounds}
        aux = dict(hm_idx=idx, hm_score=score, logits=logits, kp_detected=kp_detected, kp_synth=kp_synth,
                   dets=dets, rows=rows, objects=objects, H=H)
        return rec, aux


def loop_records(per_frame, fps, num_homography, frame_h, frame_w):
    """The reference loop body over a clip (cm.py:277-415) for keypoint_interval == 1 and ANY homography_interval:
    H is solved on scheduled frames (i % homography_interval == 0) or while the retry flag `compute_homography` is set
    (cm.py:350-351, 365-367) and carried forward otherwise.  per_frame: list of (keypoints dict as detect_keypoints
    returns it, objects dict as detect_objects returns it).  The LK-flow rescue of frames with < 4 key-points
    (cm.py:287-311) is NOT restated."""
    homography_interval = max(1, int(fps / max(1, num_homography)))
    res, H, compute_homography = {}, None, False
    for i, (kps, objects) in enumerate(per_frame):
        kps = dict(kps)
        if len(kps) >= 2:
            kps = host.synthesize_keypoints(kps)
        prev_keypoints = kps
        if i % homography_interval == 0 or compute_homography:
            img_pts, world_pts, used = host.select_plane_points(kps)
            if len(img_pts) < 4:
                compute_homography = True
            else:
                Hn, mask = P.find_homography_ransac(img_pts, world_pts, 5.0)
                if Hn is None:                                          # cm.py:354-357: RANSAC -> RHO -> LMEDS; both fall-backs are restated (oracle/eo_prims.c), see step() — this loop serves LMEDS
                    Hn, mask = P.find_homography(img_pts, world_pts, 4)
                if Hn is not None:
                    prev_keypoints = {k: v for k, v, m in zip(used, img_pts.tolist(), mask.flatten()) if m}
                    H, compute_homography = Hn, False
                else:
                    compute_homography = True
        res[i] = {"Coordinates": host.project_objects(objects, H), "Time": f"{i // fps // 60:02d}:{i // fps % 60:02d}",
                  "Keypoints": prev_keypoints, "Boundaries": host.boundaries(H, frame_h, frame_w)}
    return res

"""Drop-in for the hot path of the reference's ``CoordinateModel`` (eagle/models/coordinate_model.py:47-628) in the
stateless configuration: every per-frame number comes from the HIP library through the C ABI.

Same constructor keywords and method names as the reference: ``CoordinateModel(keypoint_conf=0.3, detector_conf=0.35)``,
``get_coordinates(frames, fps, ...)`` -> ``{i: {"Coordinates", "Time", "Keypoints", "Boundaries"}}`` (cm.py:415),
``detect_objects(frame)``, ``detect_keypoints(frame)``.  ``get_coordinates`` covers every key-point / homography cadence of
the reference (optical-flow propagation between detections, first-frame search, on-demand detection, calibration:
cm.py:188-416).  ``tracker=True`` keys Player / Goalkeeper entries by track id like the reference does through boxmot's BotSort
(cm.py:574-596): the library's BoT-SORT association (include/eagle.h, eagle_track_*); ``reid=True`` adds the appearance branch and
``camera_motion="ecc"`` boxmot's default camera-motion compensation (ECC on the 0.15-scale gray frame, eagle_clip_motion_ecc) — together the
configuration the reference constructs at cm.py:66-72; ``camera_motion=True`` / ``"sparse"`` is BoT-SORT's sparse-optical-flow alternative on a
fixed grid (eagle_clip_motion).  ``tracker=False`` (default) is the reference's detection-index fallback (cm.py:598-627), the
stateless configuration of SURVEY §8a."""
import numpy as np

from . import clip, lib, records, weights
from .pitch import INTERSECTION_TO_PITCH_POINTS


class CoordinateModel:
    def __init__(self, keypoint_conf: float = 0.3, detector_conf: float = 0.35, *, frame_hw=(720, 1280),
                 detector="n", det_imgsz=640, batch=8, precision="f32s", device=0, hrnet_state_dict=None,
                 detector_state_dict=None, seed=0, use_graph=None, multi_stream=None, tracker=False, camera_motion=False, detector_precision=None,
                 reid=False, reid_state_dict=None, allow_saturation=False, letterbox="rect"):
        # letterbox: "rect" = ultralytics LetterBox(auto=True), what the reference's .pt detectors run with (cm.py:56-57); "square" = auto=False, the static
        # det_imgsz x det_imgsz input of its exported ONNX detector (the CPU default, cm.py:54-55)
        self.keypoint_conf, self.detector_conf = keypoint_conf, detector_conf
        self.tracker = tracker
        if camera_motion not in (False, True, None, "sparse", "ecc"):
            raise ValueError("camera_motion must be False, True / 'sparse' or 'ecc'")
        self.camera_motion = "sparse" if camera_motion is True else (camera_motion or False)
        self.reid = bool(reid) and tracker      # appearance matching inside the tracker (the reference's BotSort has it on: cm.py:66-72)
        self._tracker_open = False           # the reference builds ONE BotSort in __init__ (cm.py:66-72): ids keep counting across get_coordinates calls
        self.batch = batch
        # detector_precision=None: the library's default — with precision="f32s" the detector runs in the exact fp32 family (boxes, confidences,
        # classes, NMS order and hence every detection-index id equal the fp32 reference arithmetic bit for bit), otherwise in `precision`
        # "mixed": split-family trunk, exact last C2f per level + Detect (EAGLE_DET_PREC_MIXED: measured, does not keep the exact family's ids; never a default)
        dp = {} if detector_precision is None else {"det_precision": lib.DET_PREC_MIXED if detector_precision == "mixed" else lib.PRECISIONS[detector_precision] + 1}
        self.handle = lib.Handle(device=device, frame_h=frame_hw[0], frame_w=frame_hw[1], det_variant=detector,
                                 det_imgsz=det_imgsz, batch=batch, letterbox=letterbox,
                                 precision=lib.PRECISIONS[precision],
                                 keypoint_conf=keypoint_conf, detector_conf=detector_conf,
                                 detector_floor=min(detector_conf, 0.15),
                                 use_graph=self._graph_mode(use_graph),         # None: the library's small-batch rule (include/eagle.h EAGLE_SMALL_BATCH); 0 / 1 / 2 pass through
                                 multi_stream=lib.AUTO if multi_stream is None else int(bool(multi_stream)),
                                 allow_saturation=int(allow_saturation), **dp)
        # the reference reads eagle/models/weights/*.pt|.pth (cm.py:55-59); none exist here -> seeded synthetic
        hs = hrnet_state_dict if hrnet_state_dict is not None else weights.make_hrnet_state_dict(seed)
        ys = detector_state_dict if detector_state_dict is not None else weights.make_yolo_state_dict(detector, seed)
        sds = [hs, ys]
        if self.reid:                           # the reference downloads osnet_x0_25_msmt17.pt at run time (cm.py:69); none here -> seeded synthetic
            from . import osnet
            sds.append(reid_state_dict if reid_state_dict is not None else osnet.make_osnet_state_dict(seed))
        weights.load_into(self.handle, sds)

    @staticmethod
    def _graph_mode(use_graph):
        """EagleConfig::use_graph: None -> EAGLE_AUTO; False / True -> 0 / 1; the integers 0, 1, 2 as they are (2 = replay only inside calls of at least three
        steps).  Round 5 mapped everything through bool(): mode 2 silently became per-step replay."""
        if use_graph is None:
            return lib.AUTO
        if isinstance(use_graph, (bool, np.bool_)):
            return int(use_graph)
        m = int(use_graph)
        if m not in (0, 1, 2):
            raise ValueError(f"use_graph must be None, a bool, or 0 / 1 / 2 (got {use_graph!r})")
        return m

    # ---- raw records ---------------------------------------------------------------------------------
    def process_records(self, frames):
        return self.handle.process(np.asarray(frames))

    # ---- reference-shaped API ------------------------------------------------------------------------
    def get_coordinates(self, frames, fps: int, num_homography: int = 1, num_keypoint_detection: int = 1,
                        verbose: bool = True, calibration: bool = False) -> dict:
        homography_interval = max(1, int(fps / max(1, num_homography)))
        keypoint_interval = max(1, int(fps / max(1, num_keypoint_detection)))
        if self.tracker:
            # before ANY motion estimate: eagle_track_open forgets the ECC estimator's carried template, so opening the tracker lazily after the
            # clip's camera motion (as until round 3) threw away the template this clip had just stored and mis-aligned the next clip's frame 0
            self._ensure_tracker_open()
        if calibration or keypoint_interval != 1:
            motion = [] if (self.tracker and self.camera_motion) else None
            recs = self.flow_records(frames, keypoint_interval, homography_interval, calibration, motion=motion)
            if self.tracker:
                self._track(recs, motion[0] if motion else None, frames)
            return {i: records.to_reference_dict(r, i, fps, own_h=bool(r["pad"][0])) for i, r in enumerate(recs)}
        recs = self.process_records(frames)
        if self.tracker:
            self._track(recs, self._clip_motion(frames) if self.camera_motion else None, frames)
        own = np.ones(len(recs), bool)
        if homography_interval > 1:
            # cm.py:333-367: H is solved on scheduled frames or while the retry flag is set, and carried otherwise.  Every
            # frame's own H is already in its record; decide in clip order whose H each frame uses, then let the GPU
            # re-project the frames that use a carried one.
            Hs = np.zeros((len(recs), 9)); flags = np.zeros(len(recs), np.uint8)
            cur, retry = None, False
            for i, r in enumerate(recs):
                attempt = i % homography_interval == 0 or retry
                if attempt and r["H_valid"]:
                    cur, retry = r["H"].copy(), False
                elif attempt:
                    retry = True
                own[i] = attempt and bool(r["H_valid"])
                if not own[i]:
                    flags[i] = 1 if cur is not None else 2
                    if cur is not None:
                        Hs[i] = cur
            if flags.any():
                recs = self.handle.reproject(np.ascontiguousarray(recs), Hs, flags)
        return {i: records.to_reference_dict(r, i, fps, own_h=bool(own[i])) for i, r in enumerate(recs)}

    def reset_tracker(self):
        """Forget every track: the next clip starts with a fresh tracker (frame counter 0, ids from 1), which is what a new
        ``CoordinateModel`` gives in the reference."""
        self._tracker_open = False
        if self.tracker:
            self._ensure_tracker_open()      # at once: the reset also drops the camera-motion estimator's template (boxmot's ECC object dies with its tracker)

    def _ensure_tracker_open(self):
        if not self._tracker_open:
            self.handle.track_open()
            self._tracker_open = True

    def reid_inputs(self, recs, high=0.5):
        """boxmot extracts appearance features for the high-confidence detections (conf > track_high_thresh) from ``frame[y1:y2, x1:x2]`` of
        the integer-truncated, frame-clipped box -> (crops [k,5], detection index per crop, crops per record)."""
        fh, fw = self.handle.cfg.frame_h, self.handle.cfg.frame_w
        crops, det, count = [], [], []
        for i, r in enumerate(recs):
            k = 0
            for j in range(int(r["n_det"])):
                d = r["det"][j]
                if not float(d["conf"]) > high:
                    continue
                x1, y1 = max(0, int(d["x1"])), max(0, int(d["y1"]))
                x2, y2 = min(fw - 1, int(d["x2"])), min(fh - 1, int(d["y2"]))
                if x2 > x1 and y2 > y1:
                    crops.append((i, x1, y1, x2, y2)); det.append(j); k += 1
            count.append(k)
        return np.asarray(crops, np.int32).reshape(-1, 5), np.asarray(det, np.int32), np.asarray(count, np.int32)

    def _track(self, recs, warps=None, frames=None):
        """One clip: track ids + smoothed boxes into the records (frame order), pitch coordinates re-projected on the GPU.  Like the
        reference's single BotSort instance (cm.py:66-72, 577) the tracker state lives as long as the model: ids keep increasing over
        successive clips and only the very first frame ever seen activates its tracks at once; ``reset_tracker()`` starts over."""
        self._ensure_tracker_open()
        if self.reid and frames is not None:
            crops, det, count = self.reid_inputs(recs)
            frames = np.ascontiguousarray(frames, np.uint8)
            d = self.handle.upload(frames)
            try:
                feats = self.handle.reid_features(d, len(frames), crops)
            finally:
                self.handle.free(d)
            self.handle.track_frames_reid(recs, feats, det, count, warps)
        else:
            self.handle.track_frames(recs, warps)
        return recs

    def _session_motion(self, n):
        """camera motions of the open clip session, by the configured estimator"""
        if self.camera_motion == "ecc":
            return self.handle.clip_motion_ecc(0, n, carry=True)
        return self.handle.clip_motion(0, n)

    def _clip_motion(self, frames):
        """[n, 6] camera motions of a clip that is not in a clip session yet: gray pyramids + sparse LK on the GPU (eagle_clip_open / _motion)."""
        frames = np.ascontiguousarray(frames, np.uint8)
        d = self.handle.upload(frames)
        try:
            self.handle.clip_open(d, len(frames))
            try:
                return self._session_motion(len(frames))
            finally:
                self.handle.clip_close()
        finally:
            self.handle.free(d)

    def flow_records(self, frames, keypoint_interval, homography_interval, calibration=False, stats=None, keypoint_source=None, motion=None):
        """Records of the reference loop in a stateful cadence (cm.py:188-416); see eagle_amd/clip.py."""
        frames = np.ascontiguousarray(frames, np.uint8)
        if len(frames) == 0:
            return np.zeros(0, lib.RESULT_DTYPE)
        d = self.handle.upload(frames)
        try:
            return clip.run_clip(self.handle, d, len(frames), keypoint_interval, homography_interval, calibration, stats, keypoint_source, motion,
                                 motion_fn=self._session_motion)
        finally:
            self.handle.free(d)

    def detect_objects(self, frame):
        rec = self.process_records(frame[None])[0]
        res = {"Player": {}, "Goalkeeper": {}}
        for cname, objs in records._objects(rec).items():
            for oid, d in objs.items():
                res.setdefault(cname, {})[oid] = {"BBox": [int(d["bx1"]), int(d["by1"]), int(d["bx2"]), int(d["by2"])],
                                                  "Confidence": float(d["conf"]),
                                                  "Bottom_center": [int(d["foot_x"]), int(d["foot_y"])]}
        if any(int(d["cls"]) == 2 for d in rec["det"][: int(rec["n_det"])]):
            res.setdefault("Ball", {})
        return res

    def detect_keypoints(self, frame):
        rec = self.process_records(frame[None])[0]
        return {INTERSECTION_TO_PITCH_POINTS[int(k["label"])]: (int(k["x"]), int(k["y"]))
                for k in rec["kp"][: int(rec["n_kp"])] if not k["synthesized"]}

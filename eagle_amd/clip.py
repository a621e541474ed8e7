"""Host sequencing of the reference loop in a stateful key-point cadence (eagle/models/coordinate_model.py:188-416 with
keypoint_interval > 1 and/or calibration) over the clip session of the C ABI (include/eagle.h, eagle_clip_*).

What runs where: the gray pyramids of all frames; per chunk of frames the detector on every frame and HRNet on every
keypoint_interval-th frame, in batches, on two streams; and on a third stream ONE pyramidal-LK launch and ONE loop-body launch per
frame, stream-ordered, without host round trips; the network passes of chunk c+1 run under the loop of chunk c.  The host steps in only where the reference itself leaves its cadence: the first-frame search (cm.py:289-311) and on-demand detections
(cm.py:317), after which the loop resumes at the frame that asked."""
import numpy as np

from . import lib


def run_clip(h, dptr, n, keypoint_interval, homography_interval, calibration=False, stats=None, keypoint_source=None, motion=None, motion_fn=None):
    """h: lib.Handle; dptr: n BGR frames resident in HBM.  -> RESULT_DTYPE[n].
    keypoint_source(i) -> FLOWKP_DTYPE array: an external key-point detector replacing HRNet (what detect_keypoints(frames[i])
    would return, in dict order); the parity tests use it to replay the reference's canned detections.
    motion: a list that receives the [n, 6] camera motions of the clip while the session is open (tracker CMC): motion_fn(n) if given
    (CoordinateModel's configured estimator), else eagle_clip_motion."""
    detected = []

    def detect(first, stride=1, count=1):
        idx = [first + k * stride for k in range(count)]
        if keypoint_source is None:
            h.clip_detect_keypoints(first, stride, count)
        else:
            for i in idx:
                h.clip_set_keypoints(i, keypoint_source(i))
        detected.extend(idx)

    try:
        h.clip_open(dptr, n)
        # chunks of batch * keypoint_interval frames: one full HRNet batch of scheduled frames per chunk.  Everything below is
        # enqueued asynchronously; the detector and HRNet passes of a chunk run concurrently with each other.
        chunk = max(1, int(h.cfg.batch)) * keypoint_interval
        for c0 in range(0, n, chunk):
            c1 = min(n, c0 + chunk)
            h.clip_detect_objects(c0, c1 - c0)
            detect(c0, keypoint_interval, (c1 - c0 + keypoint_interval - 1) // keypoint_interval)       # cm.py:217-276
            if c0 == 0:
                m0 = h.clip_get_keypoints(0)
                if len(m0) < 4 and n > 1:
                    _first_frame_search(h, n, m0, detect)
            h.clip_run(c0, c1, keypoint_interval, homography_interval, calibration, wait=False)
        stalled = h.clip_run(n, n, keypoint_interval, homography_interval, calibration, wait=True)
        while stalled >= 0:
            detect(stalled)                                                                          # cm.py:317 on-demand detection
            stalled = h.clip_run(stalled, n, keypoint_interval, homography_interval, calibration, wait=True)
        recs = h.clip_fetch(n)
        if motion is not None:
            motion.append(motion_fn(n) if motion_fn is not None else h.clip_motion(0, n))
    finally:
        h.clip_close()
    if stats is not None:
        stats["detected_frames"] = sorted(set(detected))
    return recs


def _first_frame_search(h, n, m0, detect):
    """cm.py:289-311: frame 0 detected fewer than 4 key-points -> find the first later frame with at least 4 and flow its
    key-points back to frame 0 (the reference tracks from gray[j] to gray[j+1] with the points of frame j+1), merging into
    mem[j] on the way.  Dict bookkeeping on the host; every flow runs on the GPU (eagle_clip_flow)."""
    prev, found = None, None
    for j in range(1, n):
        mj = h.clip_get_keypoints(j)
        if mj is None:
            detect(j)
            mj = h.clip_get_keypoints(j)
        if len(mj) >= 4:
            prev, found = mj, j
            break
    if prev is None:
        return
    for j in range(found - 1, -1, -1):
        flowed = h.clip_flow(j, j + 1, j, prev)
        prev = flowed if len(flowed) > 0 else prev
        old = h.clip_get_keypoints(j)
        merged = {int(k["label"]): k for k in prev}                             # mem[j] = {**prev_keypoints, **mem.get(j, {})}
        if old is not None:
            for k in old:
                merged[int(k["label"])] = k
        out = list(merged.values())
        if j == 0:                                                              # cm.py:324 {**keypoints, **mem[0]}: the detected keys lead
            lead = [int(k["label"]) for k in m0]
            out = [merged[l] for l in lead] + [k for l, k in merged.items() if l not in lead]
        h.clip_set_keypoints(j, np.array(out, lib.FLOWKP_DTYPE))

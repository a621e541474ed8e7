"""ORACLE — test infrastructure only (see eo_prims.c header).

BoT-SORT track association as the reference runs it behind ``self.tracker.update(dets, frame)``
(eagle/models/coordinate_model.py:66-72, 574-596; boxmot 15.0.2 ``BotSort``, uv.lock:98-99 — NOT in /root/reference and absent from
this image: restated from the published algorithm (Aharon et al. 2022; ByteTrack's two-stage association), PARITY UNPINNED).

Appearance (``with_reid=True``, the reference's configuration): ``update(..., feats=)`` takes the OSNet embeddings of the high-confidence
detections (oracle/reid.py) and fuses them as BoT-SORT does.  Camera-motion compensation: ``update(..., warp=)`` applies the
2 x 3 warp of the previous frame -> this one as BoT-SORT's ``multi_gmc`` does (means and covariances of all pooled and unconfirmed tracks, after the
prediction step); boxmot's default estimator (ECC) is oracle/ecc.py, and ``camera_motion`` below is BoT-SORT's "sparseOptFlow" alternative with a
fixed 8 x 6 grid instead of a corner detector (pyramidal LK of the key-point cadence, RANSAC over point pairs + least squares on the consensus
set).  The association itself, as published:

  * constant-velocity Kalman filter on (cx, cy, w, h) with BoT-SORT's noise scaling (std_weight_position 1/20, std_weight_velocity 1/160);
  * detections split by confidence: high (> track_high_thresh 0.5) and low (track_low_thresh 0.1 < c < 0.5);
  * 1st association: tracked + lost tracks vs high detections, cost 1 - IoU, linear assignment with cost limit match_thresh 0.8;
  * 2nd association: the still-unmatched TRACKED tracks vs low detections, cost limit 0.5; unmatched ones become lost;
  * unconfirmed tracks (born on the previous frame) vs the remaining high detections, cost 1 - IoU * detection confidence (fuse_score),
    cost limit 0.7; unmatched ones are removed;
  * remaining high detections above new_track_thresh 0.6 start tracks (activated at once only on the first frame);
  * lost tracks are removed after track_buffer 30 frames; duplicate tracked / lost pairs (IoU distance < 0.15) keep the older track;
  * output: the activated tracks in tracked state: (x1, y1, x2, y2 of the filter's posterior, id, conf, cls, detection index).

``objects_from_tracks`` is cm.py:577-596 applied to that output (the reference's own code path; IDs = track ids)."""
import numpy as np
from scipy.optimize import linear_sum_assignment

NEW, TRACKED, LOST, REMOVED = 0, 1, 2, 3
STD_POS, STD_VEL = 1.0 / 20, 1.0 / 160


class _KF:
    """KalmanFilterXYWH: state (cx, cy, w, h, vcx, vcy, vw, vh), float64."""
    F = np.eye(8)
    for _i in range(4):
        F[_i, 4 + _i] = 1.0
    H = np.eye(4, 8)

    @staticmethod
    def initiate(z):
        mean = np.r_[z, np.zeros(4)]
        std = [2 * STD_POS * z[2], 2 * STD_POS * z[3], 2 * STD_POS * z[2], 2 * STD_POS * z[3],
               10 * STD_VEL * z[2], 10 * STD_VEL * z[3], 10 * STD_VEL * z[2], 10 * STD_VEL * z[3]]
        return mean, np.diag(np.square(std))

    @staticmethod
    def predict(mean, cov):
        std = [STD_POS * mean[2], STD_POS * mean[3], STD_POS * mean[2], STD_POS * mean[3],
               STD_VEL * mean[2], STD_VEL * mean[3], STD_VEL * mean[2], STD_VEL * mean[3]]
        return _KF.F @ mean, _KF.F @ cov @ _KF.F.T + np.diag(np.square(std))

    @staticmethod
    def update(mean, cov, z):
        std = [STD_POS * mean[2], STD_POS * mean[3], STD_POS * mean[2], STD_POS * mean[3]]
        pm = _KF.H @ mean
        S = _KF.H @ cov @ _KF.H.T + np.diag(np.square(std))
        K = np.linalg.solve(S, (cov @ _KF.H.T).T).T
        return mean + K @ (z - pm), cov - K @ S @ K.T


GRID_W, GRID_H = 8, 6          # camera-motion grid (48 points: one LK launch of the key-point kernel)


def motion_grid(frame_h, frame_w):
    """centres of an 8 x 6 cell grid, truncated to whole pixels (the flow operator takes integer key-points), float32 (x, y) rows in row-major cell order"""
    return np.array([[np.floor((i + 0.5) * frame_w / GRID_W), np.floor((j + 0.5) * frame_h / GRID_H)] for j in range(GRID_H) for i in range(GRID_W)], np.float32)


def similarity_ransac(p0, p1, iters=200, thresh=3.0):
    """2 x 3 similarity [[a, -b, tx], [b, a, ty]] mapping p0 -> p1 (float64): 200 two-point hypotheses drawn by a fixed LCG, the one with the
    most residuals < 3 px wins (first on ties), least squares on its consensus set.  Every sum runs in index order (the C++ side repeats it
    operation by operation).  Fewer than two pairs, or no hypothesis: identity."""
    p0 = np.asarray(p0, np.float64).reshape(-1, 2); p1 = np.asarray(p1, np.float64).reshape(-1, 2)
    n = len(p0)
    ident = np.array([[1.0, 0.0, 0.0], [0.0, 1.0, 0.0]])
    if n < 2:
        return ident
    state = 12345
    best_cnt, best = -1, None
    for _ in range(iters):
        state = (state * 1103515245 + 12345) & 0x7FFFFFFF; i = (state >> 8) % n
        state = (state * 1103515245 + 12345) & 0x7FFFFFFF; j = (state >> 8) % n
        if i == j:
            continue
        dx0, dy0 = p0[j, 0] - p0[i, 0], p0[j, 1] - p0[i, 1]
        dx1, dy1 = p1[j, 0] - p1[i, 0], p1[j, 1] - p1[i, 1]
        den = dx0 * dx0 + dy0 * dy0
        if den < 1e-9:
            continue
        a = (dx0 * dx1 + dy0 * dy1) / den; b = (dx0 * dy1 - dy0 * dx1) / den
        tx = p1[i, 0] - (a * p0[i, 0] - b * p0[i, 1]); ty = p1[i, 1] - (b * p0[i, 0] + a * p0[i, 1])
        cnt = 0
        for k in range(n):
            ex = a * p0[k, 0] - b * p0[k, 1] + tx - p1[k, 0]; ey = b * p0[k, 0] + a * p0[k, 1] + ty - p1[k, 1]
            cnt += (ex * ex + ey * ey) < thresh * thresh
        if cnt > best_cnt:
            best_cnt, best = cnt, (a, b, tx, ty)
    if best is None:
        return ident
    a, b, tx, ty = best
    idx = [k for k in range(n) if (a * p0[k, 0] - b * p0[k, 1] + tx - p1[k, 0]) ** 2 + (b * p0[k, 0] + a * p0[k, 1] + ty - p1[k, 1]) ** 2 < thresh * thresh]
    if len(idx) < 2:
        return ident
    m = float(len(idx))
    c0x = c0y = c1x = c1y = 0.0
    for k in idx:
        c0x += p0[k, 0]; c0y += p0[k, 1]; c1x += p1[k, 0]; c1y += p1[k, 1]
    c0x /= m; c0y /= m; c1x /= m; c1y /= m
    sxx = sdot = scross = 0.0
    for k in idx:
        qx, qy, rx, ry = p0[k, 0] - c0x, p0[k, 1] - c0y, p1[k, 0] - c1x, p1[k, 1] - c1y
        sxx += qx * qx + qy * qy; sdot += qx * rx + qy * ry; scross += qx * ry - qy * rx
    if sxx < 1e-9:
        return ident
    a, b = sdot / sxx, scross / sxx
    return np.array([[a, -b, c1x - (a * c0x - b * c0y)], [b, a, c1y - (b * c0x + a * c0y)]])


def camera_motion(prev_bgr, cur_bgr):
    """the warp of frame t-1 -> t: LK (cm.py's parameters) on the grid, pairs with status 1 -> similarity_ransac"""
    from . import prims as P
    g0, g1 = P.bgr2gray(prev_bgr), P.bgr2gray(cur_bgr)
    pts = motion_grid(*g0.shape)
    nxt, st = P.calc_optical_flow_pyr_lk(g0, g1, pts)
    ok = st[:, 0] == 1
    return similarity_ransac(pts[ok], nxt[ok])


def apply_warp(tracks, warp):
    """BoT-SORT's STrack.multi_gmc: R8 = kron(I4, R) on the 8-state (it rotates (w, h) and the velocities too), t added to (cx, cy)"""
    if warp is None or not tracks:
        return
    R = np.asarray(warp, np.float64)[:, :2]; t = np.asarray(warp, np.float64)[:, 2]
    R8 = np.kron(np.eye(4), R)
    for tr in tracks:
        tr.mean = R8 @ tr.mean
        tr.mean[:2] += t
        tr.cov = R8 @ tr.cov @ R8.T


class _Track:
    def __init__(self, det, ind):
        x1, y1, x2, y2, conf, cls = [float(v) for v in det]
        self.z = np.array([(x1 + x2) / 2, (y1 + y2) / 2, x2 - x1, y2 - y1])
        self.conf, self.cls, self.det_ind = conf, int(cls), ind
        self.mean = self.cov = None
        self.state, self.is_activated, self.id = NEW, False, -1
        self.frame_id = self.start_frame = 0
        self.curr_feat = self.smooth_feat = None

    def update_features(self, feat):
        """STrack.update_features: exponential moving average (alpha 0.9) of the normalised embeddings, re-normalised"""
        if feat is None:
            return
        self.curr_feat = feat
        self.smooth_feat = feat.copy() if self.smooth_feat is None else 0.9 * self.smooth_feat + 0.1 * feat
        n = np.linalg.norm(self.smooth_feat)
        if n > 0:
            self.smooth_feat = self.smooth_feat / n

    def xyxy(self):
        c = self.z if self.mean is None else self.mean[:4]
        return np.array([c[0] - c[2] / 2, c[1] - c[3] / 2, c[0] + c[2] / 2, c[1] + c[3] / 2])


def _iou_cost(tracks, dets):
    c = np.ones((len(tracks), len(dets)))
    for i, t in enumerate(tracks):
        a = t.xyxy()
        for j, d in enumerate(dets):
            b = d.xyxy()
            iw = min(a[2], b[2]) - max(a[0], b[0]); ih = min(a[3], b[3]) - max(a[1], b[1])
            if iw > 0 and ih > 0:
                inter = iw * ih
                c[i, j] = 1.0 - inter / ((a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter)
    return c


def _fuse_appearance(cost, iou_raw, tracks, dets):
    """BoT-SORT with_reid: emb = cosine distance / 2; emb > appearance_thresh 0.25 -> 1; IoU distance > proximity_thresh 0.5 -> emb = 1;
    cost = min(cost, emb) for every pair with both features."""
    out = cost.copy()
    for i, t in enumerate(tracks):
        for j, d in enumerate(dets):
            if t.smooth_feat is None or d.curr_feat is None:
                continue
            u, v = t.smooth_feat, d.curr_feat
            e = max(0.0, 1.0 - float(u @ v) / (np.sqrt(float(u @ u)) * np.sqrt(float(v @ v)))) / 2.0
            if e > 0.25 or iou_raw[i, j] > 0.5:
                e = 1.0
            out[i, j] = min(out[i, j], e)
    return out


def _assign(cost, thresh):
    """lap.lapjv(cost, extend_cost=True, cost_limit=thresh): minimum-cost matching in which leaving a row and a column unmatched costs
    thresh (thresh / 2 each) — solved on the extended square matrix."""
    n, m = cost.shape
    if n == 0 or m == 0:
        return [], list(range(n)), list(range(m))
    ext = np.full((n + m, n + m), thresh / 2.0)
    ext[n:, m:] = 0.0
    ext[:n, :m] = cost
    r, c = linear_sum_assignment(ext)
    matches = [(int(i), int(j)) for i, j in zip(r, c) if i < n and j < m]
    mi, mj = {i for i, _ in matches}, {j for _, j in matches}
    return matches, [i for i in range(n) if i not in mi], [j for j in range(m) if j not in mj]


class BotSortLite:
    def __init__(self, track_high_thresh=0.5, track_low_thresh=0.1, new_track_thresh=0.6, track_buffer=30, match_thresh=0.8, frame_rate=30):
        self.hi, self.lo, self.new, self.match = track_high_thresh, track_low_thresh, new_track_thresh, match_thresh
        self.max_time_lost = int(frame_rate / 30.0 * track_buffer)
        self.frame_id, self.next_id = 0, 1
        self.tracked, self.lost, self.removed = [], [], []

    def _activate(self, t):
        t.mean, t.cov = _KF.initiate(t.z)
        t.update_features(t.curr_feat)
        t.id = self.next_id; self.next_id += 1
        t.state, t.is_activated = TRACKED, self.frame_id == 1
        t.frame_id = t.start_frame = self.frame_id

    def _update(self, t, d, reactivate):
        t.mean, t.cov = _KF.update(t.mean, t.cov, d.z)
        t.update_features(d.curr_feat)
        t.state, t.is_activated, t.frame_id = TRACKED, True, self.frame_id
        t.conf, t.cls, t.det_ind = d.conf, d.cls, d.det_ind

    def update(self, dets, warp=None, feats=None):
        """dets: [n,6] x1,y1,x2,y2,conf,cls -> [m,8] x1,y1,x2,y2,id,conf,cls,det_ind (as boxmot returns it).  warp: optional 2 x 3 camera
        motion of the previous frame -> this one (applied after the prediction, like BoT-SORT's gmc).  feats: optional {detection index:
        512-d embedding} of the high-confidence detections (with_reid, the reference's configuration)."""
        self.frame_id += 1
        dets = np.asarray(dets, np.float64).reshape(-1, 6)
        first = [_Track(d, i) for i, d in enumerate(dets) if d[4] > self.hi]
        if feats is not None:
            for t in first:
                f = feats.get(t.det_ind)
                if f is not None:
                    f = np.asarray(f, np.float64)
                    n = np.linalg.norm(f)
                    t.curr_feat = f / n if n > 0 else f
        second = [_Track(d, i) for i, d in enumerate(dets) if self.lo < d[4] < self.hi]
        unconfirmed = [t for t in self.tracked if not t.is_activated]
        tracked = [t for t in self.tracked if t.is_activated]
        pool = tracked + [t for t in self.lost if t not in tracked]
        for t in pool:
            if t.state != TRACKED:
                t.mean[6] = 0.0; t.mean[7] = 0.0
            t.mean, t.cov = _KF.predict(t.mean, t.cov)
        apply_warp(pool, warp)
        apply_warp(unconfirmed, warp)
        activated, refind, lost_now, removed = [], [], [], []
        c1 = _iou_cost(pool, first)
        if feats is not None:
            c1 = _fuse_appearance(c1, c1, pool, first)
        m, ut, ud = _assign(c1, self.match)
        for i, j in m:
            t = pool[i]
            (activated if t.state == TRACKED else refind).append(t)
            self._update(t, first[j], t.state != TRACKED)
        r_tracked = [pool[i] for i in ut if pool[i].state == TRACKED]
        m2, ut2, _ = _assign(_iou_cost(r_tracked, second), 0.5)
        for i, j in m2:
            t = r_tracked[i]
            (activated if t.state == TRACKED else refind).append(t)
            self._update(t, second[j], False)
        for i in ut2:
            r_tracked[i].state = LOST
            lost_now.append(r_tracked[i])
        rest = [first[j] for j in ud]
        c3 = _iou_cost(unconfirmed, rest)
        c3_raw = c3
        if c3.size:                                                        # boxmot's fuse_score: 1 - IoU * detection confidence
            c3 = 1.0 - (1.0 - c3) * np.array([d.conf for d in rest])[None, :]
            if feats is not None:
                c3 = _fuse_appearance(c3, c3_raw, unconfirmed, rest)
        m3, uu, ud3 = _assign(c3, 0.7)
        for i, j in m3:
            self._update(unconfirmed[i], rest[j], False)
            activated.append(unconfirmed[i])
        for i in uu:
            unconfirmed[i].state = REMOVED
            removed.append(unconfirmed[i])
        for j in ud3:
            if rest[j].conf >= self.new:
                self._activate(rest[j])
                activated.append(rest[j])
        for t in self.lost:
            if self.frame_id - t.frame_id > self.max_time_lost:
                t.state = REMOVED
                removed.append(t)
        self.tracked = [t for t in self.tracked if t.state == TRACKED]
        for t in activated + refind:
            if t not in self.tracked:
                self.tracked.append(t)
        self.lost = [t for t in self.lost if t not in self.tracked]
        self.lost += lost_now
        self.lost = [t for t in self.lost if t not in self.removed]      # (the published order: tracks removed on THIS frame leave the list on the next one)
        self.removed += removed
        # duplicates: a tracked and a lost track on the same object -> the older one stays
        c = _iou_cost(self.tracked, self.lost)
        da, db = set(), set()
        for i, j in zip(*np.where(c < 0.15)):
            ta, tb = self.tracked[i], self.lost[j]
            if ta.frame_id - ta.start_frame > tb.frame_id - tb.start_frame:
                db.add(j)
            else:
                da.add(i)
        self.tracked = [t for i, t in enumerate(self.tracked) if i not in da]
        self.lost = [t for j, t in enumerate(self.lost) if j not in db]
        out = [np.r_[t.xyxy(), t.id, t.conf, t.cls, t.det_ind] for t in self.tracked if t.is_activated]
        return np.array(out, np.float64).reshape(-1, 8)


def objects_from_tracks(tracks, frame_h, frame_w, detector_conf=0.35):
    """cm.py:579-596: the track rows -> {"Player": {id: {...}}, "Goalkeeper": {...}}."""
    res = {"Player": {}, "Goalkeeper": {}}
    names = {0: "Player", 1: "Goalkeeper", 2: "Ball", 3: "Referee", 4: "Staff members"}
    for x1, y1, x2, y2, tid, conf, cls, _ in tracks:
        x1 = int(np.clip(x1, 0, frame_w - 1)); y1 = int(np.clip(y1, 0, frame_h - 1))
        x2 = int(np.clip(x2, 0, frame_w - 1)); y2 = int(np.clip(y2, 0, frame_h - 1))
        label = names.get(int(cls))
        if label not in res or float(conf) < detector_conf:
            continue
        res[label][int(tid)] = {"BBox": [x1, y1, x2, y2], "Confidence": float(conf), "Bottom_center": [int((x1 + x2) / 2), y2]}
    return res

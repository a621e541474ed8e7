"""ORACLE — test infrastructure only (see eo_prims.c header).

Team colours: ``Processor.get_team_mapping`` / ``detect_color`` of the reference's post-processor (eagle/processor.py:405-503,
colour table :10-23).  K-means: ``kmeans2_labels`` restates scikit-learn 1.7's ``KMeans(n_clusters=2, random_state=0).fit(X).labels_`` (the
reference's call) step by step and is pinned to sklearn's own labels (tests/test_oracle_colors.py, several hundred crops; sklearn is in this
image); cv2's 8-bit BGR2HSV is eo_flow.c's restatement; inRange / bitwise_and / countNonZero are their numpy definitions.
Pinned to the reference's own functions by tests/golden/team_golden.json (tests/golden/make_golden.py::dump_team runs proc.py itself
over these primitives)."""
from collections import Counter

import numpy as np

from . import prims as P

COLOR_RANGES = {   # proc.py:10-23 (HSV, hue 0..180), inclusive bounds
    "red": [(0, 100, 100), (10, 255, 255)], "red2": [(160, 100, 100), (179, 255, 255)], "orange": [(11, 100, 100), (25, 255, 255)],
    "yellow": [(26, 100, 100), (35, 255, 255)], "green": [(36, 100, 100), (85, 255, 255)], "cyan": [(86, 100, 100), (95, 255, 255)],
    "blue": [(96, 100, 100), (125, 255, 255)], "purple": [(126, 100, 100), (145, 255, 255)], "magenta": [(146, 100, 100), (159, 255, 255)],
    "white": [(0, 0, 200), (180, 30, 255)], "gray": [(0, 0, 50), (180, 30, 200)], "black": [(0, 0, 0), (180, 255, 50)],
}
COLOR_ORDER = [c for c in COLOR_RANGES if c != "red2"]


def color_counts(hsv, mask):
    """pixels of `mask` inside each range -> {colour: count} with red2 merged into red (proc.py:487-498)"""
    cnt = {}
    for color, (lo, hi) in COLOR_RANGES.items():
        m = np.all((hsv >= np.array(lo, np.uint8)) & (hsv <= np.array(hi, np.uint8)), axis=2) & mask
        cnt[color] = int(m.sum())
    cnt["red"] += cnt.pop("red2")
    return cnt


# RandomState(0) is created anew by every KMeans(random_state=0).fit: its first three doubles are constants
_U0, _U1, _U2 = 0.5488135039273248, 0.7151893663724195, 0.6027633760716439


def kmeans2_labels(X):
    """sklearn.cluster.KMeans(n_clusters=2, random_state=0).fit(X).labels_ for integer pixels X [n,3] (sklearn 1.7: n_init = 1,
    k-means++ seeding, Lloyd, tol = 1e-4, max_iter = 300), restated:
      * seeding (_kmeans_plusplus): centre 0 = X[choice(n)] = X[floor(u0 n)]; two candidates (n_local_trials = 2 + int(log 2)) at
        searchsorted(cumsum(d0), (u1, u2) * sum(d0)) with d0 the squared distances to centre 0; the candidate with the smaller potential
        sum(min(d0, d_candidate)) becomes centre 1.  Squared distances between pixels are exact integers;
      * Lloyd (_kmeans_single_lloyd, float64): label = argmin_k |c_k|^2 - 2 x.c_k (ties to cluster 0), centres = cluster means; stop when the
        labels repeat, or when the summed squared centre shift is <= mean(var(X, axis=0)) * 1e-4; the returned labels are the assignment to the
        final centres.  Label 0 is the cluster that grew from centre 0."""
    X = np.asarray(X).astype(np.int64)
    n = len(X)
    i0 = min(int(np.floor(_U0 * n)), n - 1)
    d0 = ((X - X[i0]) ** 2).sum(1)
    pot = int(d0.sum())
    cum = np.cumsum(d0)
    cands = [min(int(np.searchsorted(cum, u * pot, side="left")), n - 1) for u in (_U1, _U2)]
    pots = [int(np.minimum(d0, ((X - X[c]) ** 2).sum(1)).sum()) for c in cands]
    C = np.stack([X[i0], X[cands[int(np.argmin(pots))]]]).astype(np.float64)
    Xf = X.astype(np.float64)
    mean = Xf.sum(0) / n
    tol = float(((Xf ** 2).sum(0) / n - mean * mean).sum() / 3.0 * 1e-4)
    labels, strict = None, False
    for _ in range(300):
        new = np.argmin((C ** 2).sum(1)[None, :] - 2.0 * (Xf[:, :1] * C[None, :, 0] + Xf[:, 1:2] * C[None, :, 1] + Xf[:, 2:3] * C[None, :, 2]), 1)
        Cn = C.copy()
        for k in range(2):
            m = new == k
            if m.any():
                Cn[k] = Xf[m].sum(0) / m.sum()
        same = labels is not None and np.array_equal(new, labels)
        shift = float(((Cn - C) ** 2).sum())
        labels, C = new, Cn
        if same:
            strict = True
            break
        if shift <= tol:
            break
    if not strict:
        labels = np.argmin((C ** 2).sum(1)[None, :] - 2.0 * (Xf[:, :1] * C[None, :, 0] + Xf[:, 1:2] * C[None, :, 1] + Xf[:, 2:3] * C[None, :, 2]), 1)
    return labels


def detect_color(image, labels=None):
    """proc.py:466-503 -> [(colour, count)] sorted by count, descending (stable).  labels: optional precomputed 2-means labels."""
    if labels is None:
        labels = kmeans2_labels(image[..., ::-1].reshape(-1, 3))
    labels = np.asarray(labels).reshape(image.shape[:2])
    corners = [labels[0, 0], labels[0, -1], labels[-1, 0], labels[-1, -1]]
    non_player = max(set(corners), key=corners.count)
    mask = labels == (1 if non_player == 0 else 0)
    cnt = color_counts(P.bgr2hsv(np.ascontiguousarray(image)), mask)
    return sorted([(c, n) for c, n in cnt.items() if n > 0], key=lambda x: x[1], reverse=True)


def overlap_weight(bbox, crops):
    """proc.py:420-436: the largest overlap with another player's box as a fraction of this box (crops equal to bbox are skipped)."""
    x1, y1, x2, y2 = bbox
    size = (x2 - x1) * (y2 - y1)
    mx = 0
    for c in crops:
        if c == bbox:
            continue
        ox = max(0, min(x2, c[2]) - max(x1, c[0])); oy = max(0, min(y2, c[3]) - max(y1, c[1]))
        mx = max(mx, ox * oy)
    return mx / size


def team_mapping_from_counts(per_item):
    """proc.py:405-464 after the per-crop colour lists are known.  per_item: iterable of (player_id, prop_overlap, [(colour, count)])
    in the reference's visiting order (frames in order, players in dict order); crops with prop_overlap > 0.35 never got here."""
    counts = {}
    for pid, prop, indiv in per_item:
        d = counts.setdefault(pid, {})
        for color, _ in indiv:
            d[color] = d.get(color, 0) + 1 - prop
    out = {pid: max(cc, key=cc.get) for pid, cc in counts.items()}
    most = Counter(out.values()).most_common(2)
    id_map = {c: i for i, (c, _) in enumerate(most)}
    mapping = {}
    for pid, color in out.items():
        if color in id_map:
            mapping[pid] = id_map[color]
        else:
            cc = sorted([(c, n) for c, n in counts[pid].items() if c in id_map], key=lambda x: x[1], reverse=True)
            if cc:
                mapping[pid] = id_map[cc[0][0]]
    return mapping


def get_team_mapping(frames, coords):
    items = []
    for frame, key in zip(frames, coords):
        players = coords[key].get("Coordinates", {}).get("Player", {})
        if not players:
            continue
        crops = [it["BBox"] for it in players.values()]
        for pid, it in players.items():
            prop = overlap_weight(it["BBox"], crops)
            if prop > 0.35:
                continue
            x1, y1, x2, y2 = it["BBox"]
            items.append((int(pid), prop, detect_color(frame[y1:y2, x1:x2])))
    return team_mapping_from_counts(items)

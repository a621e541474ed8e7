"""ORACLE — test infrastructure only (see eo_prims.c header).

Team colours: ``Processor.get_team_mapping`` / ``detect_color`` of the reference's post-processor (eagle/processor.py:405-503,
colour table :10-23).  K-means is scikit-learn's own ``KMeans(n_clusters=2, random_state=0)`` (the reference's call, sklearn is in this
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


def detect_color(image, labels=None):
    """proc.py:466-503 -> [(colour, count)] sorted by count, descending (stable).  labels: optional precomputed 2-means labels."""
    if labels is None:
        from sklearn.cluster import KMeans
        rgb = image[..., ::-1]
        labels = KMeans(n_clusters=2, random_state=0).fit(rgb.reshape(-1, 3)).labels_
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

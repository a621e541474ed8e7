"""Team colours for a clip (second half): the reference's ``Processor.get_team_mapping``
(eagle/processor.py:405-464) with the slow part — ``detect_color`` on every player crop (proc.py:466-503: 2-means segmentation, HSV
colour-range counts) — on the GPU over the clip that is already resident in HBM (include/eagle.h, eagle_team_colors; K15).
The host keeps what is bookkeeping in the reference too: which crops overlap another player by more than 35 %, the per-player colour
votes weighted by 1 - overlap, and the two-most-common-colours rule."""
from collections import Counter

import numpy as np

COLORS = ["red", "orange", "yellow", "green", "cyan", "blue", "purple", "magenta", "white", "gray", "black"]   # output order of the kernel (red2 merged)


def _overlap(bbox, crops):
    x1, y1, x2, y2 = bbox
    size = (x2 - x1) * (y2 - y1)
    mx = 0
    for c in crops:
        if c == bbox:
            continue
        ox = max(0, min(x2, c[2]) - max(x1, c[0])); oy = max(0, min(y2, c[3]) - max(y1, c[1]))
        mx = max(mx, ox * oy)
    return mx / size


def crop_colors(handle, dptr, n_frames, crops):
    """crops: [(frame, x1, y1, x2, y2)] -> per crop the reference's ``detect_color`` list [(colour, count)], counts descending (stable)."""
    counts = handle.team_colors(dptr, n_frames, np.asarray(crops, np.int32).reshape(-1, 5))
    out = []
    for row in counts:
        out.append(sorted([(c, int(n)) for c, n in zip(COLORS, row[:11]) if n > 0], key=lambda x: x[1], reverse=True))
    return out


def get_team_mapping(handle, dptr, coords, n_frames=None):
    """coords: ``{i: {"Coordinates": {"Player": {id: {"BBox": [x1,y1,x2,y2], ...}}}}}`` as ``get_coordinates`` returns it, frame i of the
    clip at ``dptr`` -> {player_id: 0 | 1}.  ``n_frames``: frames resident at ``dptr``; like the reference's ``zip(frames, coords)``
    (proc.py:408) the walk stops at the shorter of the two, and the kernel never indexes beyond the uploaded clip."""
    n_frames = len(coords) if n_frames is None else min(int(n_frames), len(coords))
    items, crops = [], []
    for i, key in enumerate(coords):
        if i >= n_frames:
            break
        players = coords[key].get("Coordinates", {}).get("Player", {})
        if not players:
            continue
        boxes = [[int(v) for v in it["BBox"]] for it in players.values()]
        for pid, it in players.items():
            bbox = [int(v) for v in it["BBox"]]
            if (bbox[2] - bbox[0]) * (bbox[3] - bbox[1]) <= 0:
                continue
            prop = _overlap(bbox, boxes)
            if prop > 0.35:
                continue
            items.append((int(pid), prop))
            crops.append((i, *bbox))
    if not crops:
        return {}
    colors = crop_colors(handle, dptr, n_frames, crops)
    counts = {}
    for (pid, prop), indiv in zip(items, colors):
        d = counts.setdefault(pid, {})
        for color, _ in indiv:
            d[color] = d.get(color, 0) + 1 - prop
    counts = {p: c for p, c in counts.items() if c}
    out = {pid: max(cc, key=cc.get) for pid, cc in counts.items()}
    most = Counter(out.values()).most_common(2)
    id_map = {c: k for k, (c, _) in enumerate(most)}
    mapping = {}
    for pid, color in out.items():
        if color in id_map:
            mapping[pid] = id_map[color]
        else:                                   # an outlier colour: the better of the two team colours in this player's own votes
            cc = sorted([(c, n) for c, n in counts[pid].items() if c in id_map], key=lambda x: x[1], reverse=True)
            if cc:
                mapping[pid] = id_map[cc[0][0]]
    return mapping

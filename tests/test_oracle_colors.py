"""CPU: the team-colour restatement (oracle/colors.py) against the reference's OWN Processor.get_team_mapping / detect_color outputs
(tests/golden/team_golden.json, produced by tests/golden/make_golden.py::dump_team running eagle/processor.py)."""
import json
import os

import team_cases
from oracle import colors

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "team_golden.json")))


def test_kmeans_restatement_equals_sklearn():
    """oracle/colors.py::kmeans2_labels against scikit-learn's own KMeans(n_clusters=2, random_state=0) (the reference's call, proc.py:474):
    identical label arrays (including which cluster is 0) on every golden crop, on crops of other synthetic clips, and on noise /
    two-patch / gradient images where Lloyd stops by the tolerance rule rather than at a fixed point."""
    import numpy as np
    from sklearn.cluster import KMeans
    crops = []
    for seed, ts in ((0, (0, 5, 10, 15, 20, 25)), (1, (3, 12))):
        frames, coords = team_cases.make_case(seed=seed, ts=ts)
        for i, fr in enumerate(frames):
            for p in coords[i]["Coordinates"]["Player"].values():
                x1, y1, x2, y2 = p["BBox"]
                crops.append(fr[y1:y2, x1:x2][..., ::-1].reshape(-1, 3))
    rng = np.random.default_rng(3)
    for t in range(90):
        h, w = int(rng.integers(8, 70)), int(rng.integers(8, 50))
        if t % 3 == 0:
            img = rng.integers(0, 256, (h, w, 3))
        elif t % 3 == 1:
            img = np.zeros((h, w, 3), np.int64) + rng.integers(0, 256, 3)
            img[h // 4: 3 * h // 4, w // 4: 3 * w // 4] = rng.integers(0, 256, 3)
            img = np.clip(img + rng.normal(0, 12, img.shape), 0, 255).astype(np.int64)
        else:
            img = (np.linspace(0, 255, h * w * 3).reshape(h, w, 3) + rng.normal(0, 3, (h, w, 3))).clip(0, 255).astype(np.int64)
        crops.append(img.reshape(-1, 3).astype(np.uint8))
    bad = [i for i, rgb in enumerate(crops) if not np.array_equal(KMeans(n_clusters=2, random_state=0).fit(rgb).labels_, colors.kmeans2_labels(rgb))]
    assert len(crops) > 200 and not bad, (len(crops), bad)


def test_detect_color_equals_reference_on_every_crop():
    frames, _ = team_cases.make_case()
    for c in GOLD["crops"]:
        x1, y1, x2, y2 = c["bbox"]
        assert [[k, n] for k, n in colors.detect_color(frames[c["frame"]][y1:y2, x1:x2])] == c["colors"], c


def test_team_mapping_equals_reference():
    frames, coords = team_cases.make_case()
    m = colors.get_team_mapping(frames, coords)
    assert {str(k): v for k, v in m.items()} == GOLD["team_mapping"]
    assert len(set(m.values())) == 2 and all(m[k] == m[1] for k in m if k <= 12) and all(m[k] != m[1] for k in m if k > 12)

"""-m gpu: team colours through the C ABI (eagle_team_colors, K15) against the reference's own outputs (tests/golden/team_golden.json:
eagle/processor.py's get_team_mapping / detect_color run over scikit-learn's KMeans) on a clip resident in HBM."""
import json
import os

import numpy as np
import pytest

import team_cases
from eagle_amd import lib, teams

pytestmark = pytest.mark.gpu
GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "team_golden.json")))


def test_crop_colours_and_team_mapping_equal_reference():
    frames, coords = team_cases.make_case()
    h = lib.Handle(batch=1)
    d = h.upload(np.stack(frames))
    try:
        crops = [(c["frame"], *c["bbox"]) for c in GOLD["crops"]]
        got = teams.crop_colors(h, d, len(frames), crops)
        same = sum([[k, n] for k, n in g] == c["colors"] for g, c in zip(got, GOLD["crops"]))
        top = sum((g[0][0] if g else None) == (c["colors"][0][0] if c["colors"] else None) for g, c in zip(got, GOLD["crops"]))
        print(f"crops with identical colour counts: {same} of {len(crops)}; identical dominant colour: {top}")
        # the kernel runs scikit-learn's 2-means step by step (seeding from RandomState(0)'s first doubles, tolerance rule): every crop equal
        assert same == len(crops) and top == len(crops)
        # ... and equal to the oracle on crops of another clip and of random noise (Lloyd stops by the tolerance rule there)
        from oracle import colors
        frames2, coords2 = team_cases.make_case(seed=1, ts=(3, 12))
        rng = np.random.default_rng(5)
        noise = rng.integers(0, 256, (1, 720, 1280, 3), dtype=np.uint8)
        d2 = h.upload(np.concatenate([np.stack(frames2), noise]))
        try:
            cr = [(i, *p["BBox"]) for i in range(2) for p in coords2[i]["Coordinates"]["Player"].values()]
            cr += [(2, int(x), int(y), int(x) + int(w), int(y) + int(hh)) for x, y, w, hh in zip(rng.integers(0, 1200, 40), rng.integers(0, 640, 40), rng.integers(8, 60, 40), rng.integers(8, 70, 40))]
            got2 = teams.crop_colors(h, d2, 3, cr)
            src = frames2 + [noise[0]]
            bad = [c for c, g in zip(cr, got2) if [(k, n) for k, n in g] != colors.detect_color(src[c[0]][c[2]:c[4], c[1]:c[3]])]
            assert not bad, bad
        finally:
            h.free(d2)
        m = teams.get_team_mapping(h, d, coords)
        assert {str(k): v for k, v in m.items()} == GOLD["team_mapping"]
        # degenerate crops do not break the kernel: empty, out of frame, single colour
        z = h.team_colors(d, len(frames), [(0, 10, 10, 10, 40), (0, -5, 0, 20, 20), (9, 0, 0, 8, 8), (0, 0, 0, 16, 16)])
        assert z[:3].sum() == 0 and z[3, 11] >= 0
    finally:
        h.free(d); h.close()

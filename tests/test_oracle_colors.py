"""CPU: the team-colour restatement (oracle/colors.py) against the reference's OWN Processor.get_team_mapping / detect_color outputs
(tests/golden/team_golden.json, produced by tests/golden/make_golden.py::dump_team running eagle/processor.py)."""
import json
import os

import team_cases
from oracle import colors

GOLD = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "team_golden.json")))


def test_detect_color_equals_reference_on_every_crop():
    frames, _ = team_cases.make_case()
    for c in GOLD["crops"][::3]:
        x1, y1, x2, y2 = c["bbox"]
        assert [[k, n] for k, n in colors.detect_color(frames[c["frame"]][y1:y2, x1:x2])] == c["colors"], c


def test_team_mapping_equals_reference():
    frames, coords = team_cases.make_case()
    m = colors.get_team_mapping(frames, coords)
    assert {str(k): v for k, v in m.items()} == GOLD["team_mapping"]
    assert len(set(m.values())) == 2 and all(m[k] == m[1] for k in m if k <= 12) and all(m[k] != m[1] for k in m if k > 12)

"""Frames and player boxes for the team-colour tests: synthetic frames (eagle_amd.synth) with the ground-truth rectangles of the drawn
people, grown by a margin so that the crop corners are background (what a detector box looks like)."""
import numpy as np

from eagle_amd import synth


def make_case(seed=0, ts=(0, 5, 10, 15, 20, 25), margin=5):
    frames = [synth.frame(seed, t) for t in ts]
    coords = {}
    for i, t in enumerate(ts):
        players = {}
        for k, x0, y0, x1, y1 in synth.player_boxes(seed, t):
            if k == 24:
                continue                                   # the referee is not a Player
            b = [max(0, x0 - margin), max(0, y0 - margin), min(1279, x1 + margin), min(719, y1 + margin)]
            if b[2] - b[0] < 8 or b[3] - b[1] < 8:
                continue
            players[k + 1] = {"BBox": b, "Confidence": 0.9, "Transformed_Coordinates": None}
        coords[i] = {"Coordinates": {"Player": players, "Goalkeeper": {}}, "Time": "00:00", "Keypoints": {}, "Boundaries": [None] * 4}
    return frames, coords

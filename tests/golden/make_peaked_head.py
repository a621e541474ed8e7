"""Fixture generator (test infrastructure): a head for the synthetic HRNet state-dict that produces PEAKED heat-maps.

The seeded random head of eagle_amd.weights gives noise-like heat-maps (no dominant maximum), so an fp16-vs-oracle comparison of
arg-max indices measures how often two near-equal noise values swap, not whether the fp16 family is accurate enough for a trained
network.  This script designs `unnormalized_model.1.{weight,bias}` (kh.py:553-562) as 57 matched filters over the 48-channel feature
map the synthetic backbone produces for ONE design frame (synth.frame(0, 3)): channel c responds at the projected position of pitch
landmark c (eagle_amd.synth.visible_landmarks), logit(c, q) = ALPHA * (R_c(q) - 1) + L0 with R_c the normalised correlation of the
3x3x48 patch at q with the patch at the landmark.  Landmarks that are not visible (or whose filter is not unique enough) get a
constant logit of -12.  Result: geometrically consistent key-points, so the fp16 record-level test can also compare H and pitch
coordinates.  Output: tests/golden/peaked_head.npz (weight [57,48,3,3], bias [57], the design frame's expected maxima).

    python tests/golden/make_peaked_head.py        (build container; uses the oracle's torch-CPU backend, ~20 s)"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from eagle_amd import synth, weights          # noqa: E402
from oracle import host, nets                 # noqa: E402
from oracle import prims as P                 # noqa: E402

ALPHA, L0, MARGIN = 16.0, 2.0, 0.08
DESIGN = (0, 3)


def main():
    hs = weights.make_hrnet_state_dict(0)
    frame = synth.frame(*DESIGN)
    _, feat = nets.hrnet_logits(hs, host.preprocess_keypoints(frame), backend="torch", features=True)
    F = feat[0].astype(np.float64)                                    # [135,240,48]
    Hh, Ww, C = F.shape
    pad = np.zeros((Hh + 2, Ww + 2, C)); pad[1:-1, 1:-1] = F
    patches = np.stack([pad[dy:dy + Hh, dx:dx + Ww] for dy in range(3) for dx in range(3)], 2)      # [135,240,9,48]
    mu = patches.reshape(-1, 9, C).mean(0)                                                             # [9,48]
    Wt = np.zeros((57, C, 3, 3), np.float32); b = np.full(57, -12.0, np.float32)
    vis = synth.visible_landmarks(*DESIGN)
    placed = {}
    for c, (x, y) in sorted(vis.items()):
        hx, hy = int(round(x / frame.shape[1] * (Ww - 1))), int(round(y / frame.shape[0] * (Hh - 1)))
        if not (2 <= hx < Ww - 2 and 2 <= hy < Hh - 2):
            continue
        a = patches[hy, hx] - mu
        n2 = float((a * a).sum())
        if n2 < 1e-6:
            continue
        w = ALPHA * a / n2                                            # [9,48] (tap = dy*3+dx)
        bias = L0 - ALPHA - ALPHA * float((a * mu).sum()) / n2
        logit = (patches * w[None, None]).sum((2, 3)) + bias
        s = 1.0 / (1.0 + np.exp(-logit))
        flat = s.ravel(); top = int(flat.argmax())
        second = float(np.partition(flat, -2)[-2])
        if top != hy * Ww + hx or flat[top] - second < 2 * MARGIN:   # not unique enough on the design frame: leave the landmark undetected
            continue
        Wt[c] = w.reshape(3, 3, C).transpose(2, 0, 1).astype(np.float32)
        b[c] = np.float32(bias)
        placed[c] = (hy, hx)
    hs2 = dict(hs); hs2["unnormalized_model.1.weight"] = Wt; hs2["unnormalized_model.1.bias"] = b
    lg = nets.hrnet_logits(hs2, host.preprocess_keypoints(frame), backend="torch")
    sg = P.sigmoid(lg[0]).reshape(-1, 57)
    srt = np.sort(sg, 0)
    margin = srt[-1] - srt[-2]
    ok = [c for c in placed if int(sg[:, c].argmax()) == placed[c][0] * Ww + placed[c][1] and margin[c] > MARGIN]
    print(f"{len(vis)} visible landmarks, {len(placed)} filters placed, {len(ok)} verified through the float32 network (margin > {MARGIN})")
    assert len(ok) >= 12 and len(ok) == len(placed)
    np.savez_compressed(os.path.join(HERE, "peaked_head.npz"), weight=Wt, bias=b, design=np.array(DESIGN),
                        channels=np.array(sorted(placed)), positions=np.array([placed[c] for c in sorted(placed)]))


if __name__ == "__main__":
    main()

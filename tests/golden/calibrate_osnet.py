"""Offline calibration of the synthetic OSNet-x0.25 weights (eagle_amd/osnet.py), run once in the build container:

    python tests/golden/calibrate_osnet.py        ->  eagle_amd/osnet_calib.npz

Random-weight ReLU networks produce embeddings that are almost collinear (all components positive, a large common mode): cosine distances
between ANY two crops come out ~1e-6 and the appearance term of the tracker could not separate anything.  A trained network's final
BatchNorm1d centres and scales the embedding; the same is done here for the synthetic weights: the running mean / variance of ``fc.1``
are set to the per-dimension statistics of the pre-BN embedding over the player crops of a few synthetic frames (computed with the oracle's
torch forward).  The file holds those two 512-vectors; make_osnet_state_dict() applies them.  Like CLS_BIAS_TABLE in weights.py this is data
derived offline, not code the product runs."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from eagle_amd import osnet, synth  # noqa: E402
from oracle import reid  # noqa: E402


def main():
    sd = osnet.make_osnet_state_dict(0, calibrated=False)
    # neutral final BN: read the pre-BN embedding out of the oracle
    sd[osnet.PREFIX + "fc.1.weight"] = np.ones(512, np.float32); sd[osnet.PREFIX + "fc.1.bias"] = np.full(512, 1e3, np.float32)
    sd[osnet.PREFIX + "fc.1.running_mean"] = np.zeros(512, np.float32); sd[osnet.PREFIX + "fc.1.running_var"] = np.ones(512, np.float32) - np.float32(1e-5)
    crops = []
    for seed, t in ((0, 0), (0, 7), (1, 3), (2, 11)):
        f = synth.frame(seed, t)
        for _, x0, y0, x1, y1 in synth.player_boxes(seed, t):
            r = reid.crop_box((x0, y0, x1, y1), *f.shape[:2])
            if r is not None:
                crops.append(reid.prepare_crop(f, r))
    pre = reid.embed(sd, np.stack(crops)).astype(np.float64) - 1e3
    mean, var = pre.mean(0), pre.var(0)
    np.savez(os.path.join(ROOT, "eagle_amd", "osnet_calib.npz"), fc_mean=mean.astype(np.float32), fc_var=np.maximum(var, 1e-6).astype(np.float32))
    print(len(crops), "crops; pre-BN embedding mean", float(np.abs(mean).mean()), "std", float(np.sqrt(var).mean()))


if __name__ == "__main__":
    main()

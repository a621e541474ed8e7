"""Developer probe: wall time of eagle_op_find_homography on clean correspondences (RANSAC stops after a few iterations) and on
garbage (all 2000 iterations), n points.  The op includes its own allocations and copies, so read the difference."""
import sys, time
import numpy as np
sys.path.insert(0, ".")
from eagle_amd import lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(0)
world = np.stack([rng.uniform(0, 105, n), rng.uniform(0, 68, n)], 1).astype(np.float32)
Hm = synth.camera(0, 0)
img = synth.project(Hm, world.astype(np.float64)).astype(np.float32)
garbage = np.stack([rng.uniform(0, 1280, n), rng.uniform(0, 720, n)], 1).astype(np.float32)
for name, pts in (("clean", img), ("garbage", garbage)):
    lib.op_find_homography(pts, world)
    t = time.perf_counter()
    for _ in range(20):
        H, m = lib.op_find_homography(pts, world)
    print(name, "n", n, f"{(time.perf_counter() - t) / 20 * 1e3:.3f} ms per call", "H" if H is not None else "none", int(m.sum()) if m is not None else 0)

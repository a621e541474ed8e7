"""Developer probe: in-kernel phase timers of the workgroup RANSAC (a library built with -DEAGLE_DEBUG_RANSAC, EAGLE_HIP_LIB=eagle_amd/libeagle_hip_ransacdbg.so) on
geometrically meaningless correspondences (2000 iterations, what the bench's random heads give)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eagle_amd import lib
for seed, n in ((1, 9), (2, 9), (3, 12), (4, 30)):
    rng = np.random.default_rng(100 + seed)
    img = np.floor(rng.uniform(0, 1280, (n, 2))).astype(np.float32)
    world = rng.uniform(0, 105, (n, 2)).astype(np.float32)
    print("n", n, flush=True)
    lib.op_find_homography(img, world, 5.0)
    lib.op_find_homography(img, world, 5.0)

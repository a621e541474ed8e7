import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from eagle_amd import lib
rng = np.random.default_rng(0)
def garbage(n):
    return np.floor(rng.uniform(0, 1280, (n, 2))).astype(np.float32), rng.uniform(0, 105, (n, 2)).astype(np.float32)
img, world = garbage(30)
# order of kernel launches (one homography_kernel each): see tools/prof_summary or the DB
for iters in (2000, 1000, 250, 1):
    lib.op_find_homography(img, world, 5.0, iters, 10)
lib.op_find_homography(img, world, 5.0, 2000, 0)          # no LM
Ht = np.array([[0.08, 0.01, 5.0], [0.0, 0.09, -3.0], [1e-5, 2e-5, 1.0]])
src = np.floor(rng.uniform(0, 1280, (30, 2))).astype(np.float32)
p = np.c_[src, np.ones(30)] @ Ht.T
dst = (p[:, :2] / p[:, 2:]).astype(np.float32)
lib.op_find_homography(src, dst, 5.0, 2000, 10)           # consistent: one round
lib.op_find_homography(src, dst, 5.0, 2000, 0)

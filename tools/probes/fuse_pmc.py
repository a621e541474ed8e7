"""Developer probe: one fuse_sum launch of HRNet's largest fuse (135x240x48 output, three low-resolution operands, B frames) through the
operator entry, to be run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE / TCC_HIT_sum TCC_MISS_sum (one counter set per run)."""
import sys
import numpy as np
sys.path.insert(0, ".")
from eagle_amd import lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 20
rng = np.random.default_rng(0)
base = rng.standard_normal((B, 135, 240, 48), dtype=np.float32)
ups = [rng.standard_normal((B, h, w, 48), dtype=np.float32) for h, w in ((68, 120), (34, 60), (17, 30))]
y = lib.op_fuse_sum(base, ups, True, lib.PREC_F16)
alg = (2 * base.size + sum(u.size for u in ups)) * 2
print("algorithmic bytes (fp16):", alg, "reads", (base.size + sum(u.size for u in ups)) * 2, "writes", base.size * 2)

"""Developer probe: form 1 of the fused Bottleneck against form 0 on a multi-frame map with co-resident multi-item workgroups; prints where they differ."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eagle_amd import lib
rng = np.random.default_rng(3)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
cin = 256
x = np.maximum(rng.standard_normal((B, 135, 240, cin), dtype=np.float32), 0)
w1 = (rng.standard_normal((1, 1, cin, 64)) * (2.0 / cin) ** 0.5).astype(np.float32)
w2 = (rng.standard_normal((3, 3, 64, 64)) * (2.0 / 576) ** 0.5).astype(np.float32)
w3 = (rng.standard_normal((1, 1, 64, 256)) * (2.0 / 64) ** 0.5).astype(np.float32)
b1 = np.zeros(64, np.float32); b2 = np.zeros(64, np.float32); b3 = np.zeros(256, np.float32)
os.environ["EAGLE_BNECK_FORM"] = "0"
y0 = lib.op_bottleneck(x, w1, b1, w2, b2, w3, b3)
for wgs in sys.argv[2:] or ["512"]:
    os.environ["EAGLE_BNECK_FORM"] = "1"; os.environ["EAGLE_BNECK_WGS"] = wgs
    for rep in range(3):
        y1 = lib.op_bottleneck(x, w1, b1, w2, b2, w3, b3)
        bad = ~np.isfinite(y1) | (np.abs(y1 - y0) > 1e-3 * max(1.0, np.abs(y0).max()))
        print(f"wgs {wgs} rep {rep}: bad values {int(bad.sum())} of {bad.size}; nan {int(np.isnan(y1).sum())}", flush=True)
        if bad.any():
            idx = np.argwhere(bad)
            n, r, c, ch = idx[:, 0], idx[:, 1], idx[:, 2], idx[:, 3]
            print("  frames", np.unique(n)[:10], "rows", np.unique(r)[:40], "cols(min,max)", c.min(), c.max(), "chs", np.unique(ch)[:16], len(np.unique(ch)))
            t = np.unique(np.stack([n, r // 4, c // 32], 1), axis=0)
            print("  bad tiles (n, ty, tx):", len(t), t[:12].tolist())
            k = idx[:6]
            for q in k: print("   ", q.tolist(), "form0", float(y0[tuple(q)]), "form1", float(y1[tuple(q)]), "x", float(x[q[0], q[1], q[2], q[3]]))
            pix = np.unique(np.stack([n, r, c], 1), axis=0)
            print("  bad pixels:", len(pix), "rows within tile", np.unique(pix[:, 1] % 4), "cols within tile", np.unique(pix[:, 2] % 32)[:40])

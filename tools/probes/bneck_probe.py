"""Developer probe: isolated duration of the fused Bottleneck launch (csrc/bneck.hip) at the bench's device batch.
usage: python tools/probes/bneck_probe.py [batch=50] [reps=20] [cins=256,64]   ->  one line per Cin"""
import sys, os, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eagle_amd import lib

B = int(sys.argv[1]) if len(sys.argv) > 1 else 50
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
rng = np.random.default_rng(5)
cins = [int(c) for c in sys.argv[3].split(",")] if len(sys.argv) > 3 else [256, 64]
for cin in cins:
    x = np.maximum(rng.standard_normal((B, 135, 240, cin), dtype=np.float32), 0)
    w1 = (rng.standard_normal((1, 1, cin, 64)) * (2.0 / cin) ** 0.5).astype(np.float32)
    w2 = (rng.standard_normal((3, 3, 64, 64)) * (2.0 / 576) ** 0.5).astype(np.float32)
    w3 = (rng.standard_normal((1, 1, 64, 256)) * (2.0 / 64) ** 0.5).astype(np.float32)
    b1 = np.zeros(64, np.float32); b2 = np.zeros(64, np.float32); b3 = np.zeros(256, np.float32)
    res = None if cin == 256 else rng.standard_normal((B, 135, 240, 256), dtype=np.float32)
    t0 = time.time()
    y, ms = lib.op_bottleneck(x, w1, b1, w2, b2, w3, b3, res=res, reps=reps)
    px = B * 135 * 240
    gb = px * 4.0 * (cin + 256 + (0 if res is None else 256)) / 1e9
    gf = px * 2.0 * (cin * 64 + 576 * 64 + 64 * 256) / 1e9
    print(f"bneck Cin={cin} B={B}: {ms * 1e3:.1f} us per launch  algorithmic {gb:.2f} GB -> {gb / ms:.2f} TB/s, {gf / ms:.0f} TFLOP/s  (wall {time.time() - t0:.1f} s, |y| max {np.abs(y).max():.3f})", flush=True)
    del x, y, res

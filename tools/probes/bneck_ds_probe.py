"""Developer probe: block 0 of layer 1 with the synthetic network's own (BatchNorm-folded) weights: downsample branch inside the launch against the residual-tensor path."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eagle_amd import lib, weights
hs = weights.make_hrnet_state_dict(0)
P = "unnormalized_model.0.layer1.0."
def fold(conv, bn, eps=1e-5):
    w = hs[P + conv + ".weight"].astype(np.float64)      # [co, ci, k, k]
    g, b, m, v = (hs[P + bn + s].astype(np.float64) for s in (".weight", ".bias", ".running_mean", ".running_var"))
    sc = g / np.sqrt(v + eps)
    return (w * sc[:, None, None, None]).transpose(2, 3, 1, 0).astype(np.float32), (b - m * sc).astype(np.float32)
w1, b1 = fold("conv1", "bn1"); w2, b2 = fold("conv2", "bn2"); w3, b3 = fold("conv3", "bn3"); wd, bd = fold("downsample.0", "downsample.1")
for nm, w in (("w1", w1), ("w2", w2), ("w3", w3), ("wd", wd)): print(nm, w.shape, "absmax %.4g" % np.abs(w).max(), "rms %.4g" % np.sqrt((w.astype(np.float64) ** 2).mean()))
print("b3 absmax %.4g bd absmax %.4g" % (np.abs(b3).max(), np.abs(bd).max()))
rng = np.random.default_rng(1)
x = np.maximum(rng.standard_normal((2, 40, 70, 64), dtype=np.float32), 0)
from oracle import prims as Pr
res = Pr.conv2d(x, wd, bd, stride=1, pre=0, r1=None, r2=None, post=0)
a = lib.op_bottleneck(x, w1, b1, w2, b2, w3, b3, res=res)
b = lib.op_bottleneck(x, w1, b1, w2, b2, w3, b3, wd=wd, bd=bd)
print("max |y| %.4g  max diff %.4g  rel %.3g" % (np.abs(a).max(), np.abs(a - b).max(), np.abs(a - b).max() / np.abs(a).max()))
for shp in ((3, 135, 240), (6, 135, 240)):
    x = np.maximum(rng.standard_normal(shp + (64,), dtype=np.float32), 0)
    res = Pr.conv2d(x, wd, bd, stride=1, pre=0, r1=None, r2=None, post=0)
    a = lib.op_bottleneck(x, w1, b1, w2, b2, w3, b3, res=res)
    for rep in range(3):
        b = lib.op_bottleneck(x, w1, b1, w2, b2, w3, b3, wd=wd, bd=bd)
        bad = np.abs(a - b) > 1e-4 * np.abs(a).max()
        print(shp, "rep", rep, "max |y| %.4g  max diff %.4g  bad %d" % (np.abs(a).max(), np.abs(a - b).max(), int(bad.sum())), flush=True)
        if bad.any():
            idx = np.argwhere(bad)
            print("  frames", np.unique(idx[:, 0]), "rows%8", np.unique(idx[:, 1] % 8), "cols%32", np.unique(idx[:, 2] % 32)[:20], "chs", np.unique(idx[:, 3])[:20], len(np.unique(idx[:, 3])))
# --- the real stem output of a synthetic frame as x
from eagle_amd import synth
from oracle import host
P0 = "unnormalized_model.0."
def fold0(conv, bn, eps=1e-5):
    w = hs[P0 + conv + ".weight"].astype(np.float64)
    g, b_, m, v = (hs[P0 + bn + s].astype(np.float64) for s in (".weight", ".bias", ".running_mean", ".running_var"))
    sc = g / np.sqrt(v + eps)
    return (w * sc[:, None, None, None]).transpose(2, 3, 1, 0).astype(np.float32), (b_ - m * sc).astype(np.float32)
ws1, bs1 = fold0("conv1", "bn1"); ws2, bs2 = fold0("conv2", "bn2")
xin = np.concatenate([host.preprocess_keypoints(synth.frame(0, t)) for t in range(3)])
s1 = Pr.conv2d(xin, ws1, bs1, stride=2, pre=0, r1=None, r2=None, post=1)
xs = Pr.conv2d(s1, ws2, bs2, stride=2, pre=0, r1=None, r2=None, post=1)
print("stem output", xs.shape, "max %.4g mean %.4g zeros %.3f" % (xs.max(), xs.mean(), float((xs == 0).mean())))
res = Pr.conv2d(xs, wd, bd, stride=1, pre=0, r1=None, r2=None, post=0)
a = lib.op_bottleneck(xs, w1, b1, w2, b2, w3, b3, res=res)
for rep in range(4):
    b = lib.op_bottleneck(xs, w1, b1, w2, b2, w3, b3, wd=wd, bd=bd)
    bad = np.abs(a - b) > 1e-4 * np.abs(a).max()
    print("stem x rep", rep, "max |y| %.4g  max diff %.4g  bad %d" % (np.abs(a).max(), np.abs(a - b).max(), int(bad.sum())), flush=True)
    if bad.any():
        idx = np.argwhere(bad)
        print("  frames", np.unique(idx[:, 0]), "rows%8", np.unique(idx[:, 1] % 8), "cols%32", np.unique(idx[:, 2] % 32)[:32], "chs", np.unique(idx[:, 3])[:24], len(np.unique(idx[:, 3])))
        k = idx[:5]
        for q_ in k: print("   ", q_.tolist(), float(a[tuple(q_)]), float(b[tuple(q_)]))
# --- poison LDS / registers with another kernel first (a convolution over NaNs), then the launch under test
poison_x = np.full((4, 135, 240, 64), np.nan, np.float32)
poison_w = np.full((3, 3, 64, 64), np.nan, np.float32)
for mode in ("res", "ds"):
    for rep in range(4):
        lib.op_conv2d(poison_x, poison_w, np.zeros(64, np.float32), 2, 0, None, None, 1, lib.PREC_F32S)
        b = lib.op_bottleneck(xs, w1, b1, w2, b2, w3, b3, res=res) if mode == "res" else lib.op_bottleneck(xs, w1, b1, w2, b2, w3, b3, wd=wd, bd=bd)
        bad = ~np.isfinite(b) | (np.abs(a - b) > 1e-4 * np.abs(a).max())
        print("after a poisoning launch,", mode, "rep", rep, "bad %d nan %d" % (int(bad.sum()), int(np.isnan(b).sum())), flush=True)
        if bad.any():
            idx = np.argwhere(bad)
            print("  frames", np.unique(idx[:, 0]), "rows%8", np.unique(idx[:, 1] % 8), "cols%32", np.unique(idx[:, 2] % 32)[:32], "chs", np.unique(idx[:, 3])[:24], len(np.unique(idx[:, 3])))

"""ORACLE — test infrastructure only (see eo_prims.c header).

CPU restatements of the two networks on the per-frame path, driven by a state-dict named like the reference's:

* :func:`hrnet_logits` — ``KeypointModel.forward_unnormalized`` (eagle/models/keypoint_hrnet.py:444-481,
  283-309, 553-573): HRNet-W48 + 3x3 head.  Pinned against the reference module itself
  (tests/golden/make_golden.py -> tests/golden/hrnet_*.npz).
* :func:`yolo_heads` — YOLOv8 detect forward as ultralytics 8.3.184 runs it behind
  ``self.detector_model(frame, ...)`` (eagle/models/coordinate_model.py:568); ultralytics is absent here, so
  this is restated from SURVEY App. B.1-B.2 and is PARITY UNPINNED (topology pinned by parameter counts).

Two backends compute the same graph:
  ``CBackend``     NHWC numpy + eo_prims.c: the exact-order fp32 fmaf chain the HIP fp32 kernels reproduce
                   bit-for-bit; ``f16=True`` emulates fp16 storage at exactly the points where the HIP fp16
                   path stores a tensor (the fusion plan below mirrors the product's — stated in DESIGN.md).
  ``TorchBackend`` NCHW torch-CPU fp32 (MKLDNN order): fast; used to pin the C backend against the reference
                   module and as bench.py's timed ``cpu_baseline``.
"""
import math

import numpy as np

from . import prims as P

HR = "unnormalized_model.0."
CH = (48, 96, 192, 384)


# ----------------------------------------------------------------------------------------------------------
# BN folding (eval mode): spec shared with eagle_amd/csrc/weights.cpp
#   scale = gamma / sqrt(var + eps)  in float64;  w' = f32(f64(w) * scale);  b' = f32(beta - mean * scale)
# ----------------------------------------------------------------------------------------------------------
def fold(sd, conv_name, bn_name, eps):
    w = np.asarray(sd[conv_name + ".weight"], np.float32)
    if bn_name is None:
        b = np.asarray(sd[conv_name + ".bias"], np.float32)
        return np.ascontiguousarray(w.transpose(2, 3, 1, 0)), b.copy()
    g = np.asarray(sd[bn_name + ".weight"], np.float64)
    beta = np.asarray(sd[bn_name + ".bias"], np.float64)
    mu = np.asarray(sd[bn_name + ".running_mean"], np.float64)
    var = np.asarray(sd[bn_name + ".running_var"], np.float64)
    scale = g / np.sqrt(var + np.float64(eps))
    wf = (w.astype(np.float64) * scale[:, None, None, None]).astype(np.float32)
    bf = (beta - mu * scale).astype(np.float32)
    return np.ascontiguousarray(wf.transpose(2, 3, 1, 0)), bf


class Params:
    """Folded parameters, looked up by conv module name; bn name derived by the network code."""

    def __init__(self, sd, eps, f16=False):
        self.sd, self.eps, self.f16, self._c = sd, eps, f16, {}

    def get(self, conv, bn):
        k = conv
        if k not in self._c:
            w, b = fold(self.sd, conv, bn, self.eps)
            if self.f16:
                w = P.round_f16(w)
            self._c[k] = (w, b)
        return self._c[k]


# ----------------------------------------------------------------------------------------------------------
# backends
# ----------------------------------------------------------------------------------------------------------
class CBackend:
    def __init__(self, f16=False):
        self.f16 = f16

    def input(self, x_nhwc):
        x = np.ascontiguousarray(x_nhwc, np.float32)
        return P.round_f16(x) if self.f16 else x

    def conv(self, x, wb, stride=1, pre=0, r1=None, r2=None, post=0, f32_out=False):
        w, b = wb
        return P.conv2d(x, w, b, stride=stride, pre=pre, r1=r1, r2=r2, post=post, f16_out=self.f16 and not f32_out)

    def fuse_sum(self, base, ups, relu):
        """relu(((base + up(z0)) + up(z1)) + ...) in fp32, one store."""
        N, H, W, _ = base.shape
        y = base
        for z in ups:
            y = y + P.upsample_bilinear_ac(z, H, W)
        if relu:
            y = np.maximum(y, np.float32(0))
        return P.round_f16(y) if self.f16 else y

    def cat(self, xs):
        return np.concatenate(xs, axis=3)

    def chunk2(self, x):
        c = x.shape[3] // 2
        return np.ascontiguousarray(x[..., :c]), np.ascontiguousarray(x[..., c:])

    def maxpool5(self, x):
        N, H, W, Cc = x.shape
        p = np.full((N, H + 4, W + 4, Cc), -np.inf, np.float32)
        p[:, 2:-2, 2:-2] = x
        y = x.copy()
        for dy in range(5):
            for dx in range(5):
                np.maximum(y, p[:, dy:dy + H, dx:dx + W], out=y)
        return y

    def up2(self, x):
        return np.ascontiguousarray(x.repeat(2, axis=1).repeat(2, axis=2))

    def to_nhwc(self, x):
        return x


class TorchBackend:
    def __init__(self):
        import torch
        import torch.nn.functional as F
        self.t, self.F = torch, F
        self._wc = {}

    def input(self, x_nhwc):
        return self.t.from_numpy(np.ascontiguousarray(x_nhwc, np.float32)).permute(0, 3, 1, 2).contiguous()

    def _act(self, v, a):
        if a == P.ACT_RELU:
            return self.F.relu(v)
        if a == P.ACT_SILU:
            return self.F.silu(v)
        return v

    def conv(self, x, wb, stride=1, pre=0, r1=None, r2=None, post=0, f32_out=False):
        w, b = wb
        key = id(w)
        if key not in self._wc:
            self._wc[key] = (self.t.from_numpy(np.ascontiguousarray(w.transpose(3, 2, 0, 1))), self.t.from_numpy(b))
        wt, bt = self._wc[key]
        v = self.F.conv2d(x, wt, bt, stride=stride, padding=w.shape[0] // 2)
        v = self._act(v, pre)
        if r1 is not None:
            v = r1 + v
        if r2 is not None:
            v = v + r2
        return self._act(v, post)

    def fuse_sum(self, base, ups, relu):
        y = base
        for z in ups:
            y = y + self.F.interpolate(z, size=list(base.shape[-2:]), mode="bilinear", align_corners=True)
        return self.F.relu(y) if relu else y

    def cat(self, xs):
        return self.t.cat(xs, 1)

    def chunk2(self, x):
        return x.chunk(2, 1)

    def maxpool5(self, x):
        return self.F.max_pool2d(x, 5, 1, 2)

    def up2(self, x):
        return self.F.interpolate(x, scale_factor=2, mode="nearest")

    def to_nhwc(self, x):
        return x.permute(0, 2, 3, 1).contiguous().numpy()


# ----------------------------------------------------------------------------------------------------------
# HRNet-W48 + head
# ----------------------------------------------------------------------------------------------------------
def _hr_stage(be, prm, xs, stage_idx, n_modules, nb, last_single):
    R = P.ACT_RELU
    for m in range(n_modules):
        q = f"{HR}stage{stage_idx}.{m}."
        for b in range(nb):                                    # branches: 4 BasicBlocks each (kh.py:83-99)
            x = xs[b]
            for k in range(4):
                r = f"{q}branches.{b}.{k}."
                o = be.conv(x, prm.get(r + "conv1", r + "bn1"), post=R)
                x = be.conv(o, prm.get(r + "conv2", r + "bn2"), r1=x, post=R)
            xs[b] = x
        n_out = 1 if (last_single and m == n_modules - 1) else nb
        out = []
        for i in range(n_out):                                 # fuse (kh.py:290-309): left-assoc sum, then ReLU
            y = None
            for j in range(i):                                 # j < i : chain of (i-j) 3x3 s2 convs
                t = xs[j]
                for k in range(i - j):
                    r = f"{q}fuse_layers.{i}.{j}.{k}."
                    last = k == i - j - 1
                    if not last:
                        t = be.conv(t, prm.get(r + "0", r + "1"), stride=2, post=R)
                    else:
                        ident = xs[i] if j == i - 1 else None          # the j == i term follows immediately
                        relu_now = ident is not None and i == nb - 1
                        t = be.conv(t, prm.get(r + "0", r + "1"), stride=2, r1=y, r2=ident, post=R if relu_now else 0)
                y = t
            if i == 0:
                y = xs[0]
            ups = []
            for j in range(i + 1, nb):                         # j > i : 1x1 conv + BN, bilinear up (align_corners)
                r = f"{q}fuse_layers.{i}.{j}."
                ups.append(be.conv(xs[j], prm.get(r + "0", r + "1")))
            if ups:
                y = be.fuse_sum(y, ups, True)
            out.append(y)
        xs = out
    return xs


def hrnet_logits(sd, x_nhwc, backend="c", f16=False, prm=None, features=False):
    """x [N,540,960,3] normalised RGB -> logits [N,135,240,57] float32 (before the sigmoid).
    features=True: also the 48-channel map the head reads ([N,135,240,48]; used by tests/golden/make_peaked_head.py)."""
    be = CBackend(f16) if backend == "c" else TorchBackend()
    prm = prm or Params(sd, 1e-5, f16 and backend == "c")
    R = P.ACT_RELU
    x = be.input(x_nhwc)
    x = be.conv(x, prm.get(HR + "conv1", HR + "bn1"), stride=2, post=R)
    x = be.conv(x, prm.get(HR + "conv2", HR + "bn2"), stride=2, post=R)
    for b in range(4):                                         # layer1: Bottlenecks (kh.py:117-137)
        q = f"{HR}layer1.{b}."
        res = be.conv(x, prm.get(q + "downsample.0", q + "downsample.1")) if b == 0 else x
        o = be.conv(x, prm.get(q + "conv1", q + "bn1"), post=R)
        o = be.conv(o, prm.get(q + "conv2", q + "bn2"), post=R)
        x = be.conv(o, prm.get(q + "conv3", q + "bn3"), r1=res, post=R)
    x0 = be.conv(x, prm.get(HR + "transition1.0.0", HR + "transition1.0.1"), post=R)
    x1 = be.conv(x, prm.get(HR + "transition1.1.0.0", HR + "transition1.1.0.1"), stride=2, post=R)
    ys = _hr_stage(be, prm, [x0, x1], 2, 1, 2, False)
    ys.append(be.conv(ys[-1], prm.get(HR + "transition2.2.0.0", HR + "transition2.2.0.1"), stride=2, post=R))
    ys = _hr_stage(be, prm, ys, 3, 4, 3, False)
    ys.append(be.conv(ys[-1], prm.get(HR + "transition3.3.0.0", HR + "transition3.3.0.1"), stride=2, post=R))
    ys = _hr_stage(be, prm, ys, 4, 3, 4, True)
    logits = be.conv(ys[0], prm.get("unnormalized_model.1", None), f32_out=True)
    if features:
        return be.to_nhwc(logits), np.asarray(be.to_nhwc(ys[0]), np.float32)
    return be.to_nhwc(logits)


# ----------------------------------------------------------------------------------------------------------
# YOLOv8 detect
# ----------------------------------------------------------------------------------------------------------
def yolo_heads(sd, x_nhwc, variant="n", backend="c", f16=False, prm=None):
    """x [N,H,W,3] RGB/255 letterboxed -> list of 3 (box_logits [N,h,w,64], cls_logits [N,h,w,nc]) per level."""
    from eagle_amd.weights import yolo_channels  # architecture constants only (depth/width table)
    be = CBackend(f16) if backend == "c" else TorchBackend()
    prm = prm or Params(sd, 1e-3, f16 and backend == "c")
    S = P.ACT_SILU
    n1, n2, n3, n4 = yolo_channels(variant)["n"]

    def cv(x, name, stride=1, r1=None):
        return be.conv(x, prm.get(name + ".conv", name + ".bn"), stride=stride, pre=S, r1=r1)

    def c2f(x, i, n, shortcut):
        y = list(be.chunk2(cv(x, f"model.{i}.cv1")))
        for k in range(n):
            t = cv(y[-1], f"model.{i}.m.{k}.cv1")
            y.append(cv(t, f"model.{i}.m.{k}.cv2", r1=y[-1] if shortcut else None))
        return cv(be.cat(y), f"model.{i}.cv2")

    x = be.input(x_nhwc)
    x = cv(x, "model.0", 2)
    x = cv(x, "model.1", 2)
    x = c2f(x, 2, n1, True)
    x = cv(x, "model.3", 2)
    p3 = c2f(x, 4, n2, True)
    x = cv(p3, "model.5", 2)
    p4 = c2f(x, 6, n3, True)
    x = cv(p4, "model.7", 2)
    x = c2f(x, 8, n4, True)
    x = cv(x, "model.9.cv1")
    y1 = be.maxpool5(x); y2 = be.maxpool5(y1); y3 = be.maxpool5(y2)
    p5 = cv(be.cat([x, y1, y2, y3]), "model.9.cv2")
    h12 = c2f(be.cat([be.up2(p5), p4]), 12, n1, False)
    h15 = c2f(be.cat([be.up2(h12), p3]), 15, n1, False)
    h18 = c2f(be.cat([cv(h15, "model.16", 2), h12]), 18, n1, False)
    h21 = c2f(be.cat([cv(h18, "model.19", 2), p5]), 21, n1, False)
    outs = []
    for l, f in enumerate((h15, h18, h21)):
        b = cv(cv(f, f"model.22.cv2.{l}.0"), f"model.22.cv2.{l}.1")
        b = be.conv(b, prm.get(f"model.22.cv2.{l}.2", None), f32_out=True)
        c = cv(cv(f, f"model.22.cv3.{l}.0"), f"model.22.cv3.{l}.1")
        c = be.conv(c, prm.get(f"model.22.cv3.{l}.2", None), f32_out=True)
        outs.append((be.to_nhwc(b), be.to_nhwc(c)))
    return outs


def yolo_decode(heads, nc=5):
    """Detect inference tail (App. B.2): concat levels -> [A, 4+nc] rows (cx,cy,w,h in input pixels, class probs)."""
    rows = []
    for (b, c), s in zip(heads, (8.0, 16.0, 32.0)):
        assert b.shape[0] == 1
        rows.append(P.yolo_decode_level(b[0], c[0], nc, s))
    return np.concatenate(rows, 0)

"""OSNet-x0.25 (Zhou et al., "Omni-Scale Feature Learning for Person Re-Identification", ICCV 2019) as the reference's tracker uses it:
``BotSort(reid_weights=Path("osnet_x0_25_msmt17.pt"), ...)`` (eagle/models/coordinate_model.py:66-72; boxmot 15.0.2 builds torchreid's
``osnet_x0_25`` for that file name).  Neither boxmot nor the checkpoint exists here (downloaded at run time by the reference): the
architecture is restated from the publication / torchreid's module layout, weights are seeded synthetic — PARITY UNPINNED.

    conv1   ConvLayer(3, 16, 7, stride 2, pad 3) = conv + BN + ReLU      256 x 128 -> 128 x 64
    maxpool 3 x 3, stride 2, pad 1                                        -> 64 x 32
    conv2   OSBlock(16, 64), OSBlock(64, 64), Conv1x1(64, 64) + AvgPool(2)  -> 32 x 16
    conv3   OSBlock(64, 96), OSBlock(96, 96), Conv1x1(96, 96) + AvgPool(2)  -> 16 x 8
    conv4   OSBlock(96, 128), OSBlock(128, 128)
    conv5   Conv1x1(128, 128)
    global average pool -> fc: Linear(128, 512) + BatchNorm1d + ReLU -> the 512-d embedding (eval mode returns it)
    OSBlock(cin, cout), mid = cout // 4:  x1 = Conv1x1(cin, mid); four streams of 1 .. 4 LightConv3x3(mid, mid) (1x1 linear conv ->
      depthwise 3x3 -> BN -> ReLU); x2 = sum_k gate(stream_k) with ONE shared ChannelGate(mid) (global average -> 1x1 conv to mid // 16 with
      bias -> ReLU -> 1x1 conv back with bias -> sigmoid -> scale); out = ReLU(Conv1x1Linear(mid, cout)(x2) + identity), identity through
      Conv1x1Linear(cin, cout) when cin != cout.
This module enumerates the parameter tensors with torchreid's names and generates the synthetic state-dict; the forward passes live in
the HIP library (csrc/reid.hip) and, for the tests, in oracle/reid.py."""
import math
import os
from collections import OrderedDict

import numpy as np

from .weights import _rng

CHANNELS = (16, 64, 96, 128)
FEATURE_DIM = 512
CROP_H, CROP_W = 256, 128
PREFIX = "reid."


def blocks():
    """(name, cin, cout) of the six OSBlocks in forward order."""
    c = CHANNELS
    return [("conv2.0", c[0], c[1]), ("conv2.1", c[1], c[1]), ("conv3.0", c[1], c[2]), ("conv3.1", c[2], c[2]), ("conv4.0", c[2], c[3]), ("conv4.1", c[3], c[3])]


def _bn(sd, seed, name, c, gamma=1.0):
    r = _rng(seed, name)
    sd[name + ".weight"] = (gamma * r.uniform(0.8, 1.2, c)).astype(np.float32)
    sd[name + ".bias"] = (0.1 * r.standard_normal(c)).astype(np.float32)
    sd[name + ".running_mean"] = (0.1 * r.standard_normal(c)).astype(np.float32)
    sd[name + ".running_var"] = r.uniform(0.7, 1.3, c).astype(np.float32)


def _conv(sd, seed, name, cout, cin, k, gain=1.0):
    g = _rng(seed, name)
    sd[name] = (g.standard_normal((cout, cin, k, k), dtype=np.float32) * np.float32(gain * math.sqrt(2.0 / (cin * k * k))))


def make_osnet_state_dict(seed=0, calibrated=True):
    """Seeded synthetic weights for osnet_x0_25 under torchreid's parameter names, prefixed ``reid.`` (201,864 parameters without the classifier).
    calibrated: the final BatchNorm1d centres / scales the embedding with statistics measured offline over synthetic player crops
    (tests/golden/calibrate_osnet.py -> osnet_calib.npz; seed 0 only), as a trained network's would — without it a random ReLU network's
    embeddings are collinear to ~1e-6 and carry no appearance signal."""
    sd = OrderedDict()
    P = PREFIX
    _conv(sd, seed, P + "conv1.conv.weight", CHANNELS[0], 3, 7); _bn(sd, seed, P + "conv1.bn", CHANNELS[0])
    for name, cin, cout in blocks():
        b, mid = P + name, cout // 4
        _conv(sd, seed, b + ".conv1.conv.weight", mid, cin, 1); _bn(sd, seed, b + ".conv1.bn", mid)
        for s, depth in (("a", 1), ("b", 2), ("c", 3), ("d", 4)):
            for k in range(depth):
                lc = b + (".conv2a" if s == "a" else f".conv2{s}.{k}")
                _conv(sd, seed, lc + ".conv1.weight", mid, mid, 1, gain=0.8)
                g = _rng(seed, lc + ".conv2.weight")
                sd[lc + ".conv2.weight"] = (g.standard_normal((mid, 1, 3, 3), dtype=np.float32) * np.float32(math.sqrt(2.0 / 9)))      # depthwise
                _bn(sd, seed, lc + ".bn", mid)
        r = max(mid // 16, 1)
        _conv(sd, seed, b + ".gate.fc1.weight", r, mid, 1); sd[b + ".gate.fc1.bias"] = (0.1 * _rng(seed, b + ".gate.fc1.bias").standard_normal(r)).astype(np.float32)
        _conv(sd, seed, b + ".gate.fc2.weight", mid, r, 1); sd[b + ".gate.fc2.bias"] = (0.5 * _rng(seed, b + ".gate.fc2.bias").standard_normal(mid)).astype(np.float32)
        _conv(sd, seed, b + ".conv3.conv.weight", cout, mid, 1, gain=0.35); _bn(sd, seed, b + ".conv3.bn", cout)
        if cin != cout:
            _conv(sd, seed, b + ".downsample.conv.weight", cout, cin, 1, gain=0.7); _bn(sd, seed, b + ".downsample.bn", cout)
    for name, c in (("conv2.2.0", CHANNELS[1]), ("conv3.2.0", CHANNELS[2]), ("conv5", CHANNELS[3])):
        _conv(sd, seed, P + name + ".conv.weight", c, c, 1); _bn(sd, seed, P + name + ".bn", c)
    g = _rng(seed, P + "fc.0.weight")
    sd[P + "fc.0.weight"] = (g.standard_normal((FEATURE_DIM, CHANNELS[3]), dtype=np.float32) * np.float32(math.sqrt(2.0 / CHANNELS[3])))
    sd[P + "fc.0.bias"] = (0.1 * _rng(seed, P + "fc.0.bias").standard_normal(FEATURE_DIM)).astype(np.float32)
    _bn(sd, seed, P + "fc.1", FEATURE_DIM)
    calib = os.path.join(os.path.dirname(os.path.abspath(__file__)), "osnet_calib.npz")
    if calibrated and seed == 0 and os.path.exists(calib):
        c = np.load(calib)
        sd[P + "fc.1.running_mean"] = c["fc_mean"].astype(np.float32)
        sd[P + "fc.1.running_var"] = c["fc_var"].astype(np.float32)
        sd[P + "fc.1.weight"] = np.ones(FEATURE_DIM, np.float32)
        sd[P + "fc.1.bias"] = np.zeros(FEATURE_DIM, np.float32)
    return sd

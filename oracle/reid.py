"""ORACLE — test infrastructure only (see eo_prims.c header).

Appearance embeddings of the reference's tracker (eagle/models/coordinate_model.py:66-72: ``BotSort(reid_weights="osnet_x0_25_msmt17.pt")``,
fed with the BGR frame at cm.py:577).  boxmot 15.0.2 / torchreid are not in /root/reference and absent from this image: OSNet-x0.25 is
restated from the published architecture (eagle_amd/osnet.py has the layer table) over a state-dict with torchreid's parameter names, in
numpy float32 — PARITY UNPINNED.  Crop preparation as boxmot's ReID backends do it: ``frame[y1:y2, x1:x2]`` of the integer-truncated,
frame-clipped box -> cv2.resize to 128 x 256 (INTER_LINEAR; restated by eo_resize_linear_u8c3) -> BGR2RGB -> / 255 -> ImageNet mean / std."""
import numpy as np

from . import prims as P

MEAN = np.array([0.485, 0.456, 0.406], np.float32)
STD = np.array([0.229, 0.224, 0.225], np.float32)
CROP_H, CROP_W = 256, 128
PREFIX = "reid."


def crop_box(box, frame_h, frame_w):
    """integer crop rectangle of a float xyxy box (astype(int) truncation, clipped to the frame); None when empty"""
    x1, y1, x2, y2 = (int(v) for v in box[:4])
    x1, y1 = max(0, x1), max(0, y1)
    x2, y2 = min(frame_w - 1, x2), min(frame_h - 1, y2)
    return (x1, y1, x2, y2) if (x2 > x1 and y2 > y1) else None


def prepare_crop(frame_bgr, rect):
    x1, y1, x2, y2 = rect
    c = P.resize_linear_u8c3(np.ascontiguousarray(frame_bgr[y1:y2, x1:x2]), CROP_H, CROP_W)[:, :, ::-1].astype(np.float32)
    return ((c / np.float32(255.0) - MEAN) / STD).astype(np.float32)          # HWC, RGB


def _w(sd, name):
    return np.asarray(sd[PREFIX + name], np.float32)


def _bn(sd, name, x, eps=1e-5):
    """eval-mode BatchNorm over the last (channel) axis of an NHWC / NC array, float32"""
    sc = (_w(sd, name + ".weight").astype(np.float64) / np.sqrt(_w(sd, name + ".running_var").astype(np.float64) + eps))
    sh = _w(sd, name + ".bias").astype(np.float64) - _w(sd, name + ".running_mean").astype(np.float64) * sc
    return (x * sc.astype(np.float32) + sh.astype(np.float32)).astype(np.float32)


def _conv1x1(x, w):
    """x [n,h,w,cin], w [cout,cin,1,1] -> [n,h,w,cout]"""
    return np.tensordot(x, w[:, :, 0, 0].T, axes=([3], [0])).astype(np.float32)


def _conv_bn(sd, name, x, relu):
    y = _bn(sd, name + ".bn", _conv1x1(x, _w(sd, name + ".conv.weight")))
    return np.maximum(y, np.float32(0)) if relu else y


def _conv7_s2(x, w):
    """7 x 7, stride 2, pad 3: x [n,256,128,3] -> [n,128,64,cout] through an im2col view"""
    n, h, wd, c = x.shape
    xp = np.pad(x, ((0, 0), (3, 3), (3, 3), (0, 0)))
    ho, wo = h // 2, wd // 2
    s = xp.strides
    cols = np.lib.stride_tricks.as_strided(xp, (n, ho, wo, 7, 7, c), (s[0], 2 * s[1], 2 * s[2], s[1], s[2], s[3]))
    return np.tensordot(cols, w.transpose(2, 3, 1, 0), axes=([3, 4, 5], [0, 1, 2])).astype(np.float32)          # w [cout,cin,7,7] -> [7,7,cin,cout]


def _maxpool3_s2(x):
    n, h, wd, c = x.shape
    xp = np.pad(x, ((0, 0), (1, 1), (1, 1), (0, 0)), constant_values=-np.inf)
    ho, wo = (h + 2 - 3) // 2 + 1, (wd + 2 - 3) // 2 + 1
    s = xp.strides
    win = np.lib.stride_tricks.as_strided(xp, (n, ho, wo, 3, 3, c), (s[0], 2 * s[1], 2 * s[2], s[1], s[2], s[3]))
    return win.max((3, 4))


def _light(sd, name, x):
    """LightConv3x3: 1x1 linear convolution -> depthwise 3x3 (pad 1) -> BatchNorm -> ReLU"""
    y = _conv1x1(x, _w(sd, name + ".conv1.weight"))
    k = _w(sd, name + ".conv2.weight")[:, 0]                       # [c, 3, 3]
    yp = np.pad(y, ((0, 0), (1, 1), (1, 1), (0, 0)))
    h, wd = y.shape[1:3]
    acc = np.zeros_like(y)
    for ky in range(3):
        for kx in range(3):
            acc += yp[:, ky:ky + h, kx:kx + wd, :] * k[:, ky, kx]
    return np.maximum(_bn(sd, name + ".bn", acc), np.float32(0))


def _gate(sd, name, x):
    g = x.mean((1, 2), dtype=np.float32)                            # [n, c]
    g = np.maximum(g @ _w(sd, name + ".fc1.weight")[:, :, 0, 0].T + _w(sd, name + ".fc1.bias"), np.float32(0))
    g = g @ _w(sd, name + ".fc2.weight")[:, :, 0, 0].T + _w(sd, name + ".fc2.bias")
    g = (1.0 / (1.0 + np.exp(-g.astype(np.float64)))).astype(np.float32)
    return x * g[:, None, None, :]


def _osblock(sd, name, x, cin, cout):
    x1 = _conv_bn(sd, name + ".conv1", x, True)
    streams = [_light(sd, name + ".conv2a", x1)]
    for s, depth in (("b", 2), ("c", 3), ("d", 4)):
        y = x1
        for k in range(depth):
            y = _light(sd, f"{name}.conv2{s}.{k}", y)
        streams.append(y)
    x2 = ((_gate(sd, name + ".gate", streams[0]) + _gate(sd, name + ".gate", streams[1])) + _gate(sd, name + ".gate", streams[2])) + _gate(sd, name + ".gate", streams[3])
    x3 = _conv_bn(sd, name + ".conv3", x2, False)
    ident = _conv_bn(sd, name + ".downsample", x, False) if cin != cout else x
    return np.maximum(x3 + ident, np.float32(0))


def embed(sd, crops):
    """crops: float32 [n, 256, 128, 3] (prepare_crop) -> [n, 512] embeddings (OSNet in eval mode returns the fc output).  Plain numpy float32
    (no torch in the process that also drives the GPU library: a ROCm torch next to the library's dlopen'ed RCCL aborted at interpreter exit)."""
    from eagle_amd import osnet
    if len(crops) == 0:
        return np.zeros((0, 512), np.float32)
    out = []
    for i0 in range(0, len(crops), 16):                             # bounded working set
        x = np.ascontiguousarray(np.asarray(crops[i0:i0 + 16], np.float32))
        x = np.maximum(_bn(sd, "conv1.bn", _conv7_s2(x, _w(sd, "conv1.conv.weight"))), np.float32(0))
        x = _maxpool3_s2(x)
        for i, (name, cin, cout) in enumerate(osnet.blocks()):
            x = _osblock(sd, name, x, cin, cout)
            if i in (1, 3):
                t = _conv_bn(sd, name[:5] + ".2.0", x, True)
                n, h, wd, c = t.shape
                x = t.reshape(n, h // 2, 2, wd // 2, 2, c).mean((2, 4), dtype=np.float32)
        x = _conv_bn(sd, "conv5", x, True)
        v = x.mean((1, 2), dtype=np.float32)
        v = (v @ _w(sd, "fc.0.weight").T + _w(sd, "fc.0.bias")).astype(np.float32)
        out.append(np.maximum(_bn(sd, "fc.1", v), np.float32(0)))
    return np.concatenate(out)


def features(sd, frame_bgr, boxes):
    """[len(boxes), 512] embeddings of the boxes' crops (zeros for an empty crop)."""
    h, w = frame_bgr.shape[:2]
    rects = [crop_box(b, h, w) for b in boxes]
    out = np.zeros((len(boxes), 512), np.float32)
    idx = [i for i, r in enumerate(rects) if r is not None]
    if idx:
        out[idx] = embed(sd, np.stack([prepare_crop(frame_bgr, rects[i]) for i in idx]))
    return out

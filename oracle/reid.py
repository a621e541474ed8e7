"""ORACLE — test infrastructure only (see eo_prims.c header).

Appearance embeddings of the reference's tracker (eagle/models/coordinate_model.py:66-72: ``BotSort(reid_weights="osnet_x0_25_msmt17.pt")``,
fed with the BGR frame at cm.py:577).  boxmot 15.0.2 / torchreid are not in /root/reference and absent from this image: OSNet-x0.25 is
restated from the published architecture (eagle_amd/osnet.py has the layer table) over a state-dict with torchreid's parameter names, in
torch-CPU fp32 — PARITY UNPINNED.  Crop preparation as boxmot's ReID backends do it: ``frame[y1:y2, x1:x2]`` of the integer-truncated,
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


def _t(sd, name):
    import torch
    return torch.from_numpy(np.ascontiguousarray(sd[PREFIX + name]))


def _bn(sd, name, x, eps=1e-5):
    import torch.nn.functional as F
    return F.batch_norm(x, _t(sd, name + ".running_mean"), _t(sd, name + ".running_var"), _t(sd, name + ".weight"), _t(sd, name + ".bias"), False, 0.0, eps)


def _conv_bn(sd, name, x, relu, stride=1, pad=0):
    import torch.nn.functional as F
    y = _bn(sd, name + ".bn", F.conv2d(x, _t(sd, name + ".conv.weight"), None, stride, pad))
    return F.relu(y) if relu else y


def _light(sd, name, x):
    import torch.nn.functional as F
    y = F.conv2d(x, _t(sd, name + ".conv1.weight"))
    y = F.conv2d(y, _t(sd, name + ".conv2.weight"), None, 1, 1, 1, y.shape[1])
    return F.relu(_bn(sd, name + ".bn", y))


def _gate(sd, name, x):
    import torch
    import torch.nn.functional as F
    g = x.mean((2, 3), keepdim=True)
    g = F.relu(F.conv2d(g, _t(sd, name + ".fc1.weight"), _t(sd, name + ".fc1.bias")))
    g = torch.sigmoid(F.conv2d(g, _t(sd, name + ".fc2.weight"), _t(sd, name + ".fc2.bias")))
    return x * g


def _osblock(sd, name, x, cin, cout):
    import torch.nn.functional as F
    x1 = _conv_bn(sd, name + ".conv1", x, True)
    a = _light(sd, name + ".conv2a", x1)
    streams = [a]
    for s, depth in (("b", 2), ("c", 3), ("d", 4)):
        y = x1
        for k in range(depth):
            y = _light(sd, f"{name}.conv2{s}.{k}", y)
        streams.append(y)
    x2 = sum(_gate(sd, name + ".gate", y) for y in streams)
    x3 = _conv_bn(sd, name + ".conv3", x2, False)
    ident = _conv_bn(sd, name + ".downsample", x, False) if cin != cout else x
    return F.relu(x3 + ident)


def embed(sd, crops):
    """crops: float32 [n, 256, 128, 3] (prepare_crop) -> [n, 512] embeddings (OSNet in eval mode returns the fc output)."""
    import torch
    import torch.nn.functional as F
    from eagle_amd import osnet
    if len(crops) == 0:
        return np.zeros((0, 512), np.float32)
    with torch.no_grad():
        x = torch.from_numpy(np.ascontiguousarray(np.asarray(crops, np.float32).transpose(0, 3, 1, 2)))
        x = _conv_bn(sd, "conv1", x, True, 2, 3)
        x = F.max_pool2d(x, 3, 2, 1)
        for i, (name, cin, cout) in enumerate(osnet.blocks()):
            x = _osblock(sd, name, x, cin, cout)
            if i in (1, 3):
                x = F.avg_pool2d(_conv_bn(sd, name[:5] + ".2.0", x, True), 2, 2)
        x = _conv_bn(sd, "conv5", x, True)
        v = x.mean((2, 3))
        v = F.linear(v, _t(sd, "fc.0.weight"), _t(sd, "fc.0.bias"))
        v = F.batch_norm(v, _t(sd, "fc.1.running_mean"), _t(sd, "fc.1.running_var"), _t(sd, "fc.1.weight"), _t(sd, "fc.1.bias"), False, 0.0, 1e-5)
        return F.relu(v).numpy()


def features(sd, frame_bgr, boxes):
    """[len(boxes), 512] embeddings of the boxes' crops (zeros for an empty crop)."""
    h, w = frame_bgr.shape[:2]
    rects = [crop_box(b, h, w) for b in boxes]
    out = np.zeros((len(boxes), 512), np.float32)
    idx = [i for i, r in enumerate(rects) if r is not None]
    if idx:
        out[idx] = embed(sd, np.stack([prepare_crop(frame_bgr, rects[i]) for i in idx]))
    return out

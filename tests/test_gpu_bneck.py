"""-m gpu: the fused Bottleneck launch of HRNet's layer 1 (csrc/bneck.hip, round 6; eagle/models/keypoint_hrnet.py:101-137) through the C ABI's
operator entry, against (a) the fp32 oracle's three convolutions and (b) the three unfused split-family launches it replaces."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F32S_TOL = 4e-6          # one split-family convolution against the fp32 oracle (tests/test_gpu_ops.py)
BNECK_TOL = 3 * F32S_TOL  # three chained convolutions, the two intermediates rounded to the split format (22+ bits) where the oracle keeps fp32


def _rand(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


def _weights(cin, seed):
    w1 = _rand((1, 1, cin, 64), seed, (2.0 / cin) ** 0.5); b1 = _rand((64,), seed + 1, 0.1)
    w2 = _rand((3, 3, 64, 64), seed + 2, (2.0 / 576) ** 0.5); b2 = _rand((64,), seed + 3, 0.1)
    w3 = _rand((1, 1, 64, 256), seed + 4, (2.0 / 64) ** 0.5); b3 = _rand((256,), seed + 5, 0.1)
    return w1, b1, w2, b2, w3, b3


def _oracle(x, ws, res):
    from oracle import prims as P
    w1, b1, w2, b2, w3, b3 = ws
    t1 = P.conv2d(x, w1, b1, stride=1, pre=0, r1=None, r2=None, post=1)
    t2 = P.conv2d(t1, w2, b2, stride=1, pre=0, r1=None, r2=None, post=1)
    return P.conv2d(t2, w3, b3, stride=1, pre=0, r1=x if res is None else res, r2=None, post=1)


SHAPES = [(1, 8, 32), (2, 19, 45), (3, 1, 1), (1, 7, 65), (5, 17, 30), (1, 9, 33), (2, 16, 64)]


@pytest.mark.parametrize("p1", ["ring", "direct"])
@pytest.mark.parametrize("form", ["0", "1"])
@pytest.mark.parametrize("wgs", ["256", "8"])
@pytest.mark.parametrize("cin", [256, 64, 16])
@pytest.mark.parametrize("shape", SHAPES)
def test_fused_bottleneck_against_the_oracle(shape, cin, wgs, form, p1, monkeypatch):
    """Ragged maps, partial tiles in both directions, a 1 x 1 map, several tiles per workgroup (EAGLE_BNECK_WGS=8: the x ring runs on across items and the
    next tile's first chunks are requested under phases 2 / 3), the identity shortcut (Cin = 256) and a separate residual tensor (block 0: Cin = 64)."""
    from eagle_amd import lib
    monkeypatch.setenv("EAGLE_BNECK_WGS", wgs)
    monkeypatch.setenv("EAGLE_BNECK_FORM", form)        # 0: tile 8 x 32, one 8-wave workgroup per CU; 1: tile 4 x 32, two 4-wave workgroups per CU, one-slot x ring
    monkeypatch.setenv("EAGLE_BNECK_P1", p1)            # phase 1: x through the LDS-DMA ring (default) / pixel fragments straight from global memory (Cin = 256, 64)
    if p1 == "direct" and cin == 16:
        pytest.skip("the B-direct form exists for Cin = 256 and 64; any other Cin takes the ring")
    n, h, w = shape
    x = _rand((n, h, w, cin), 71)
    ws = _weights(cin, 72)
    res = None if cin == 256 else _rand((n, h, w, 256), 79)
    ref = _oracle(x, ws, res)
    got = lib.op_bottleneck(x, *ws, res=res)
    err = np.abs(ref - got).max() / max(np.abs(ref).max(), 1e-6)
    assert err < BNECK_TOL, f"fused bottleneck error {err}"


@pytest.mark.parametrize("form", ["0", "1"])
@pytest.mark.parametrize("cin", [256, 64])
def test_fused_bottleneck_full_map_against_the_unfused_launches(cin, form, monkeypatch):
    """One 135 x 240 frame (HRNet's layer-1 map: 17 x 8 tiles, the last tile row one pixel row short, the last column half empty) with 17 tiles per workgroup,
    against the three launches the kernel replaces (same family, same rounding of t1 / t2 to the split format) and against the oracle."""
    from eagle_amd import lib
    monkeypatch.setenv("EAGLE_BNECK_WGS", "8")
    monkeypatch.setenv("EAGLE_BNECK_FORM", form)
    x = np.maximum(_rand((1, 135, 240, cin), 81), 0)          # post-ReLU activations, as in the network
    ws = _weights(cin, 82)
    w1, b1, w2, b2, w3, b3 = ws
    res = None if cin == 256 else _rand((1, 135, 240, 256), 89)
    got = lib.op_bottleneck(x, *ws, res=res)
    t1 = lib.op_conv2d(x, w1, b1, 1, 0, None, None, 1, lib.PREC_F32S)
    t2 = lib.op_conv2d(t1, w2, b2, 1, 0, None, None, 1, lib.PREC_F32S)
    unf = lib.op_conv2d(t2, w3, b3, 1, 0, x if res is None else res, None, 1, lib.PREC_F32S)
    scale = max(np.abs(unf).max(), 1e-6)
    assert np.abs(unf - got).max() / scale < BNECK_TOL
    ref = _oracle(x, ws, res)
    assert np.abs(ref - got).max() / scale < BNECK_TOL


@pytest.mark.parametrize("form", ["0", "1"])
def test_fused_bottleneck_zero_padding_of_the_intermediate(form, monkeypatch):
    """conv2 pads t1 with ZEROS, not with relu(b1): a kernel that computed conv1 on the out-of-image halo and kept the result would differ on every border pixel.
    Large positive conv1 biases make that difference huge."""
    from eagle_amd import lib
    monkeypatch.setenv("EAGLE_BNECK_FORM", form)
    x = _rand((1, 11, 37, 256), 91)
    w1, b1, w2, b2, w3, b3 = _weights(256, 92)
    b1 = np.abs(b1) + 3.0
    ref = _oracle(x, (w1, b1, w2, b2, w3, b3), None)
    got = lib.op_bottleneck(x, w1, b1, w2, b2, w3, b3)
    assert np.abs(ref - got).max() / np.abs(ref).max() < BNECK_TOL


def test_fused_bottleneck_forms_agree_with_co_resident_workgroups(monkeypatch):
    """Eight 135 x 240 frames: 2176 tiles of the 4 x 32 form on 512 workgroups — TWO co-resident workgroups per CU, several tiles each — against the 8 x 32 form.  Both
    forms accumulate every output in the same order, so the results must be identical bit for bit.  This is the configuration that exposed the store-data hazard of
    csrc/bneck.hip (a VALU write to a wide store's data register in the issue slot behind it: 1 - 2 thousand corrupted values per run before the explicit wait states);
    small maps and one workgroup per CU never showed it.  Three repetitions: the corruption was timing-dependent."""
    from eagle_amd import lib
    x = np.maximum(_rand((8, 135, 240, 256), 95), 0)
    ws = _weights(256, 96)
    monkeypatch.setenv("EAGLE_BNECK_FORM", "0")
    y0 = lib.op_bottleneck(x, *ws)
    assert np.isfinite(y0).all()
    monkeypatch.setenv("EAGLE_BNECK_FORM", "1")
    for wgs, p1 in (("512", "ring"), ("1024", "ring"), ("512", "direct"), ("512", "ring")):
        monkeypatch.setenv("EAGLE_BNECK_WGS", wgs)
        monkeypatch.setenv("EAGLE_BNECK_P1", p1)
        y1 = lib.op_bottleneck(x, *ws)
        assert np.array_equal(y0, y1), f"form 1 ({wgs} workgroups, phase 1 {p1}) differs from form 0 in {int((y0 != y1).sum())} values"


@pytest.mark.parametrize("form", ["0", "1"])
@pytest.mark.parametrize("wgs", ["256", "8"])
@pytest.mark.parametrize("shape", SHAPES + [(1, 135, 240)])
def test_fused_bottleneck_with_the_downsample_branch_inside(shape, wgs, form, monkeypatch):
    """Block 0 of layer 1 (kh.py:328): the shortcut is a 1 x 1 convolution 64 -> 256 + BatchNorm of x.  With (wd, bd) the launch computes it itself — conv3 over K = 128 =
    [t2 | x], both weight sets on one power-of-two scale, bias b3 + bd, x's inner pixels as B-direct fragments — instead of reading a residual tensor: against the oracle's
    FOUR convolutions, and against the same launch fed with the oracle's downsample output as a residual tensor."""
    from eagle_amd import lib
    from oracle import prims as P
    monkeypatch.setenv("EAGLE_BNECK_WGS", wgs)
    monkeypatch.setenv("EAGLE_BNECK_FORM", form)
    n, h, w = shape
    x = np.maximum(_rand((n, h, w, 64), 101), 0)
    ws = _weights(64, 102)
    wd = _rand((1, 1, 64, 256), 108, (2.0 / 64) ** 0.5 * 0.37)      # another magnitude than W3: the common scale must serve both
    bd = _rand((256,), 109, 0.1)
    res = P.conv2d(x, wd, bd, stride=1, pre=0, r1=None, r2=None, post=0)
    ref = _oracle(x, ws, res)
    got = lib.op_bottleneck(x, *ws, wd=wd, bd=bd)
    scale = max(np.abs(ref).max(), 1e-6)
    assert np.abs(ref - got).max() / scale < BNECK_TOL + F32S_TOL, f"fused bottleneck + downsample error {np.abs(ref - got).max() / scale}"
    via_tensor = lib.op_bottleneck(x, *ws, res=res)
    assert np.abs(via_tensor - got).max() / scale < BNECK_TOL


def test_fused_bottleneck_after_a_launch_of_another_kernel(monkeypatch):
    """The second appearance of the store-data hazard (csrc/bneck.hip, phase 3): in the downsample-fused instantiation hipcc had moved the next block's VALU work in
    front of the explicit wait states, and 16 values per 0.4 G came out wrong — but only when the launch followed ANOTHER kernel (in the pipeline: the stem), never in a
    sequence of Bottleneck launches, which is why every operator-level test passed while whole frames failed.  Here a convolution over NaNs precedes each launch; both
    shortcut forms, full-size maps, against a result computed first.  (tools/isa_store_hazard.py is the static check of the same rule, tests/test_isa_lint.py.)"""
    from eagle_amd import lib
    from oracle import prims as P
    x = np.maximum(_rand((4, 135, 240, 64), 121), 0)
    ws = _weights(64, 122)
    wd = _rand((1, 1, 64, 256), 123, (2.0 / 64) ** 0.5 * 0.5)
    bd = _rand((256,), 124, 0.1)
    res = P.conv2d(x[:1], wd, bd, stride=1, pre=0, r1=None, r2=None, post=0)
    first = lib.op_bottleneck(x, *ws, wd=wd, bd=bd)
    scale = np.abs(first).max()
    assert np.abs(lib.op_bottleneck(x[:1], *ws, res=res) - first[:1]).max() / scale < BNECK_TOL
    poison_x = np.full((4, 135, 240, 64), np.nan, np.float32)
    poison_w = np.full((3, 3, 64, 64), np.nan, np.float32)
    for rep in range(6):
        lib.op_conv2d(poison_x, poison_w, np.zeros(64, np.float32), 2, 0, None, None, 1, lib.PREC_F32S)
        again = lib.op_bottleneck(x, *ws, wd=wd, bd=bd)
        assert np.array_equal(first, again), f"repetition {rep}: {int((first != again).sum())} values differ after a launch of another kernel"

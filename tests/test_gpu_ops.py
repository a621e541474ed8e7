"""-m gpu: each HIP kernel of the path against the CPU oracle, through the C ABI's operator entry points.
fp32 kernels must be BIT-EXACT (same fmaf chain as oracle/eo_prims.c); fp16 kernels within the stated tolerance."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

F16_TOL = 2e-3  # relative to max|y|: fp16 output rounding (2^-11) + reordered fp32 accumulation
# Split family (EAGLE_PREC_F32S): operands carry 22+ bits, products hi*hi + hi*lo + lo*hi are accumulated in fp32 in the MFMA's order.
# Measured on MI355X against float64 (tools/probes/split_mfma_probe.hip): rms error 2.0e-7 .. 6.5e-7 of the output rms for K = 288 .. 3456,
# i.e. that of the fp32 fmaf chain itself; the bound below is on the largest deviation from the fp32 oracle relative to max|y|.
F32S_TOL = 4e-6


def _split_round(a):
    """What storing an array in the split format keeps: hi = rn16(16 v), lo = rn16(16 v - hi) -> (hi + lo) / 16."""
    s = a.astype(np.float32) * np.float32(16)
    hi = s.astype(np.float16).astype(np.float32)
    lo = (s - hi).astype(np.float16).astype(np.float32)
    return (hi + lo) * np.float32(0.0625)


def _rand(shape, seed, scale=1.0):
    return (np.random.default_rng(seed).standard_normal(shape) * scale).astype(np.float32)


CONV_CASES = [
    # n, h, w, cin, cout, ks, stride, pre, post, r1, r2
    (1, 20, 37, 48, 48, 3, 1, 0, 1, False, False),     # HRNet branch-0 shape, ragged tile edges
    (2, 17, 30, 96, 96, 3, 1, 0, 1, True, False),      # BasicBlock conv2 with residual
    (1, 33, 61, 48, 96, 3, 2, 0, 1, True, True),       # fuse downsample with running sum + identity
    (1, 9, 13, 192, 384, 3, 2, 0, 0, False, False),
    (1, 19, 23, 64, 256, 1, 1, 0, 1, True, False),     # Bottleneck conv3
    (1, 16, 16, 384, 48, 1, 1, 0, 0, False, False),    # fuse 1x1
    (1, 45, 52, 3, 64, 3, 2, 0, 1, False, False),      # stem
    (1, 30, 41, 48, 57, 3, 1, 0, 0, False, False),     # head (57 real channels)
    (1, 24, 40, 16, 16, 3, 1, 2, 0, True, False),      # YOLO bottleneck: x + silu(conv)
    (1, 12, 20, 256, 128, 1, 1, 2, 0, False, False),   # YOLO 1x1 + SiLU
    (1, 12, 20, 64, 5, 1, 1, 0, 0, False, False),      # class head
    (3, 1, 1, 16, 16, 3, 1, 0, 0, False, False),       # degenerate 1x1 map, batch 3
]


@pytest.mark.parametrize("case", CONV_CASES)
@pytest.mark.parametrize("prec", ["f32", "f16", "f32s"])
def test_conv_parity(case, prec):
    from eagle_amd import lib
    from oracle import prims as P
    n, h, w, cin, cout, ks, st, pre, post, use_r1, use_r2 = case
    x = _rand((n, h, w, cin), 1)
    wt = _rand((ks, ks, cin, cout), 2, (2.0 / (cin * ks * ks)) ** 0.5)
    b = _rand((cout,), 3, 0.1)
    ho = (h + 2 * (ks // 2) - ks) // st + 1
    wo = (w + 2 * (ks // 2) - ks) // st + 1
    r1 = _rand((n, ho, wo, cout), 4) if use_r1 else None
    r2 = _rand((n, ho, wo, cout), 5) if use_r2 else None
    if prec == "f32":
        ref = P.conv2d(x, wt, b, stride=st, pre=pre, r1=r1, r2=r2, post=post)
        got = lib.op_conv2d(x, wt, b, st, pre, r1, r2, post, lib.PREC_F32)
        assert np.array_equal(ref, got), f"fp32 conv not bit-exact: max|d|={np.abs(ref - got).max()}"
    elif prec == "f32s":
        ref = P.conv2d(x, wt, b, stride=st, pre=pre, r1=r1, r2=r2, post=post)          # the fp32 oracle itself, no emulation
        got = lib.op_conv2d(x, wt, b, st, pre, r1, r2, post, lib.PREC_F32S)
        err = np.abs(ref - got).max() / max(np.abs(ref).max(), 1e-6)
        assert err < F32S_TOL, f"split conv error {err}"
    else:
        q = P.round_f16
        ref = P.conv2d(q(x), q(wt), b, stride=st, pre=pre, r1=None if r1 is None else q(r1),
                       r2=None if r2 is None else q(r2), post=post, f16_out=True)
        got = lib.op_conv2d(x, wt, b, st, pre, r1, r2, post, lib.PREC_F16)
        err = np.abs(ref - got).max() / max(np.abs(ref).max(), 1e-6)
        assert err < F16_TOL, f"fp16 conv error {err}"


@pytest.mark.parametrize("stack", ["1", "0"])
@pytest.mark.parametrize("force", ["3,2,0", "3,1,0", "6,2,3", "3,1,3", "2,2,4", "1,1,4"])
@pytest.mark.parametrize("shape,ks,st,cin,cout", [((5, 17, 30), 3, 1, 48, 96), ((4, 7, 45), 3, 1, 32, 48), ((3, 34, 61), 3, 2, 48, 96), ((6, 12, 20), 1, 1, 64, 96),
                                                  ((7, 1, 1), 3, 1, 16, 48), ((2, 33, 37), 3, 2, 3, 48), ((50, 17, 30), 3, 1, 32, 48)])
def test_conv_f32_every_tiling_is_bit_exact(shape, ks, st, cin, cout, force, stack, monkeypatch):
    """The exact family's tilings (round 4): full / half / quarter tiles (variants 0 / 3 / 4), 16 x 16 or 8 x 32 sub-tile arrangement, and the
    batch tiled as ONE image of N (H + 1) rows for stride-1 layers (a tile may straddle several frames; the virtual zero row between two
    frames is both frames' padding).  The K order of every output is the canonical one whatever the tiling: bit-equal to the oracle, with
    residual and ReLU, on maps whose rows do not divide the tile (17, 7, 1) and batches of 2 - 7 frames."""
    from eagle_amd import lib
    from oracle import prims as P
    monkeypatch.setenv("EAGLE_F32_FORCE", force)
    monkeypatch.setenv("EAGLE_F32_STACK", stack)
    n, h, w = shape
    x = _rand((n, h, w, cin), 61)
    wt = _rand((ks, ks, cin, cout), 62, (2.0 / (cin * ks * ks)) ** 0.5)
    b = _rand((cout,), 63, 0.1)
    ho = (h + 2 * (ks // 2) - ks) // st + 1
    wo = (w + 2 * (ks // 2) - ks) // st + 1
    r1 = _rand((n, ho, wo, cout), 64)
    ref = P.conv2d(x, wt, b, stride=st, pre=0, r1=r1, r2=None, post=1)
    got = lib.op_conv2d(x, wt, b, st, 0, r1, None, 1, lib.PREC_F32)
    assert np.array_equal(ref, got), f"fp32 conv tiling {force} stack {stack} not bit-exact: max|d|={np.abs(ref - got).max()}"


@pytest.mark.parametrize("force,cin,cout", [("48,3,6", 48, 48), ("48,3,7", 48, 48), ("96,3,7", 96, 96), ("96,3,6", 96, 96),
                                            ("64,4,7", 64, 64), ("96,2,7", 96, 32)])
@pytest.mark.parametrize("shape", [(3, 37, 45), (2, 5, 70), (1, 16, 16)])
def test_conv_weight_stationary_variants(force, cin, cout, shape, monkeypatch):
    """The persistent weight-stationary 3x3 kernels (conv.hip variants 6/7), forced through EAGLE_CONV_FORCE, against the
    oracle on ragged maps (partial tiles, more workgroup slots than tiles, several tiles per workgroup)."""
    from eagle_amd import lib
    from oracle import prims as P
    monkeypatch.setenv("EAGLE_CONV_FORCE", force)
    n, h, w = shape
    x = _rand((n, h, w, cin), 11)
    wt = _rand((3, 3, cin, cout), 12, (2.0 / (cin * 9)) ** 0.5)
    b = _rand((cout,), 13, 0.1)
    r1 = _rand((n, h, w, cout), 14)
    q = P.round_f16
    ref = P.conv2d(q(x), q(wt), b, stride=1, pre=0, r1=q(r1), r2=None, post=1, f16_out=True)
    got = lib.op_conv2d(x, wt, b, 1, 0, r1, None, 1, lib.PREC_F16)
    err = np.abs(ref - got).max() / max(np.abs(ref).max(), 1e-6)
    assert err < F16_TOL, f"fp16 weight-stationary conv error {err}"


@pytest.mark.parametrize("cin,cout", [(96, 96), (192, 192), (64, 192), (384, 384), (32, 96)])
@pytest.mark.parametrize("shape", [(3, 37, 45), (2, 5, 70), (1, 16, 16), (5, 17, 30), (1, 68, 120)])
@pytest.mark.parametrize("res,post", [(True, 1), (False, 1), (False, 0), (2, 1)])
def test_conv_a_direct_variants(cin, cout, shape, res, post):
    _a_direct_case(cin, cout, shape, res, post, 1)


@pytest.mark.parametrize("cin,cout", [(48, 96), (96, 192), (256, 96), (192, 384), (48, 192), (96, 96), (40, 96)])
@pytest.mark.parametrize("shape", [(3, 37, 45), (2, 5, 70), (1, 16, 16), (2, 135, 240), (3, 34, 61), (1, 1, 1)])
@pytest.mark.parametrize("res,post", [(True, 1), (False, 0), (2, 1), (2, 0)])
def test_conv_a_direct_stride2(cin, cout, shape, res, post):
    """The stride-2 form (variants 10 / 11: the kernel runs on the space-to-depth image, re-arranged by the LDS-DMA addressing): odd and
    even input sizes, partial tiles, Cin that is not a multiple of 32 (chunks spanning two phases), a 1 x 1 input."""
    if shape[1] * shape[2] > 4000 and (cin > 96 or not res):
        pytest.skip("large map: one representative case")
    _a_direct_case(cin, cout, shape, res, post, 2)


def _a_direct_case(cin, cout, shape, res, post, stride):
    """The A-direct 3x3 stride-1 kernels (conv.hip variants 8 / 9: weight fragments straight from global memory, activations by
    LDS-DMA), which conv_choose picks for Cout = 96 / 192 / 384, against the oracle: ragged maps (partial tiles in both directions,
    fewer items than workgroups, several items per workgroup and Cout blocks), with / without residual and ReLU."""
    from eagle_amd import lib
    from oracle import prims as P
    if stride == 1 and shape[1] * shape[2] > 4000 and (cin > 96 or not res):
        pytest.skip("large map: one representative case")
    n, h, w = shape
    ho, wo = (h + 2 - 3) // stride + 1, (w + 2 - 3) // stride + 1
    x = _rand((n, h, w, cin), 21)
    wt = _rand((3, 3, cin, cout), 22, (2.0 / (cin * 9)) ** 0.5)
    b = _rand((cout,), 23, 0.1)
    r1 = _rand((n, ho, wo, cout), 24) if res else None
    r2 = _rand((n, ho, wo, cout), 25) if res == 2 else None          # res: False / True / 2 = number of residual operands
    q = P.round_f16
    ref = P.conv2d(q(x), q(wt), b, stride=stride, pre=0, r1=q(r1) if res else None, r2=q(r2) if res == 2 else None, post=post, f16_out=True)
    got = lib.op_conv2d(x, wt, b, stride, 0, r1, r2, post, lib.PREC_F16)
    err = np.abs(ref - got).max() / max(np.abs(ref).max(), 1e-6)
    assert err < F16_TOL, f"fp16 A-direct conv error {err}"


@pytest.mark.parametrize("force", ["16,1,0", "16,2,0", "16,3,0", "16,4,0", "32,3,0", "32,4,0", "16,6,3", "32,6,3", "16,4,3", "32,2,3"])
@pytest.mark.parametrize("cin,cout,ks,stride", [(96, 96, 3, 1), (64, 192, 3, 2), (192, 96, 1, 1)])
def test_conv_split_variants(force, cin, cout, ks, stride, monkeypatch):
    """Every (kc, nt, tile) instance class of the split family through EAGLE_CONV_FORCE, on a ragged map with two residuals and ReLU,
    against the fp32 oracle."""
    from eagle_amd import lib
    from oracle import prims as P
    kc, nt, _ = (int(v) for v in force.split(","))
    if cout % (16 * nt) or cin % kc or (ks == 1 and kc == 8):
        pytest.skip("shape does not fit this instance")
    monkeypatch.setenv("EAGLE_CONV_FORCE", force)
    n, h, w = 2, 19, 45
    ho, wo = (h + 2 * (ks // 2) - ks) // stride + 1, (w + 2 * (ks // 2) - ks) // stride + 1
    x = _rand((n, h, w, cin), 31)
    wt = _rand((ks, ks, cin, cout), 32, (2.0 / (cin * ks * ks)) ** 0.5)
    b = _rand((cout,), 33, 0.1)
    r1, r2 = _rand((n, ho, wo, cout), 34), _rand((n, ho, wo, cout), 35)
    ref = P.conv2d(x, wt, b, stride=stride, pre=0, r1=r1, r2=r2, post=1)
    got = lib.op_conv2d(x, wt, b, stride, 0, r1, r2, 1, lib.PREC_F32S)
    err = np.abs(ref - got).max() / max(np.abs(ref).max(), 1e-6)
    assert err < F32S_TOL, f"split conv error {err}"


@pytest.mark.parametrize("cin,cout", [(96, 96), (192, 192), (48, 192), (384, 384), (144, 96), (48, 48), (96, 48), (48, 144)])
@pytest.mark.parametrize("shape", [(3, 37, 45), (2, 5, 70), (1, 16, 16), (5, 17, 30), (1, 68, 120)])
@pytest.mark.parametrize("res,post", [(True, 1), (False, 1), (False, 0), (2, 1)])
def test_conv_split_a_direct(cin, cout, shape, res, post, monkeypatch):
    """The 16x16x32 A-direct kernels of the split family (conv_ad_split.inc; until round 5 conv_choose's pick for 3x3 stride-1 layers with Cin = 48 k and
    Cout = 96 k; since then behind EAGLE_CONV_M32=0, the 32x32x16 forms being the default: test_conv_split_a_direct_m32) against the fp32 oracle: ragged maps,
    several items per workgroup and Cout blocks, 0 / 1 / 2 residual operands."""
    from eagle_amd import lib
    from oracle import prims as P
    monkeypatch.setenv("EAGLE_CONV_M32", "0")
    if shape[1] * shape[2] > 4000 and (cin > 96 or not res):
        pytest.skip("large map: one representative case")
    n, h, w = shape
    x = _rand((n, h, w, cin), 51)
    wt = _rand((3, 3, cin, cout), 52, (2.0 / (cin * 9)) ** 0.5)
    b = _rand((cout,), 53, 0.1)
    r1 = _rand((n, h, w, cout), 54) if res else None
    r2 = _rand((n, h, w, cout), 55) if res == 2 else None
    ref = P.conv2d(x, wt, b, stride=1, pre=0, r1=r1, r2=r2, post=post)
    got = lib.op_conv2d(x, wt, b, 1, 0, r1, r2, post, lib.PREC_F32S)
    err = np.abs(ref - got).max() / max(np.abs(ref).max(), 1e-6)
    assert err < F32S_TOL, f"split A-direct conv error {err}"


@pytest.mark.parametrize("cin,cout", [(96, 96), (192, 192), (48, 192), (384, 384), (64, 96), (16, 192), (32, 288), (144, 96)])
@pytest.mark.parametrize("shape", [(3, 37, 45), (2, 5, 70), (1, 16, 16), (5, 17, 30), (1, 68, 120), (1, 1, 1), (2, 9, 33), (2, 34, 60), (1, 7, 65)])
@pytest.mark.parametrize("res,post", [(True, 1), (False, 1), (False, 0), (2, 1)])
@pytest.mark.parametrize("tile", ["rows", "wide64"])
def test_conv_split_a_direct_m32(cin, cout, shape, res, post, tile, monkeypatch):
    """Round 5: the A-direct kernels of the split family on v_mfma_f32_32x32x16_f16 (conv_ad_split32.inc; variant 21: BN = 192, tile 4 x 32; variant 22:
    BN = 96, tile 8 x 32; wave tile 96 channels x 2 rows x 32 pixels), forced through EAGLE_CONV_FORCE, against the fp32 oracle: ragged maps, partial tiles in
    both directions (a 33-column map: one pixel in the second tile), a 1 x 1 map, several items per workgroup and Cout blocks, 0 / 1 / 2 residual operands, and
    Cin = 16 k that is NOT a multiple of 48 (this form has no ring-phase restriction)."""
    from eagle_amd import lib
    from oracle import prims as P
    if shape[1] * shape[2] > 4000 and (cin > 96 or not res):
        pytest.skip("large map: one representative case")
    # tile "rows": the wave's two 32-pixel blocks in two rows (variants 21 / 22: tiles 4 x 32 / 8 x 32); "wide64": side by side (variants 23 / 24: tiles 2 x 64 / 4 x 64)
    v = (21 if cout % 192 == 0 else 22) + (2 if tile == "wide64" else 0)
    monkeypatch.setenv("EAGLE_CONV_FORCE", f"16,{12 if cout % 192 == 0 else 6},{v}")
    n, h, w = shape
    x = _rand((n, h, w, cin), 51)
    wt = _rand((3, 3, cin, cout), 52, (2.0 / (cin * 9)) ** 0.5)
    b = _rand((cout,), 53, 0.1)
    r1 = _rand((n, h, w, cout), 54) if res else None
    r2 = _rand((n, h, w, cout), 55) if res == 2 else None
    ref = P.conv2d(x, wt, b, stride=1, pre=0, r1=r1, r2=r2, post=post)
    got = lib.op_conv2d(x, wt, b, 1, 0, r1, r2, post, lib.PREC_F32S)
    err = np.abs(ref - got).max() / max(np.abs(ref).max(), 1e-6)
    assert err < F32S_TOL, f"split A-direct 32x32x16 conv (variant {v}) error {err}"


@pytest.mark.parametrize("shape", [(3, 37, 45), (1, 16, 16), (2, 135, 240)])
@pytest.mark.parametrize("res,post", [(True, 1), (False, 0), (2, 1)])
@pytest.mark.parametrize("form", ["12", "13", "19"])
def test_conv_split_a_direct_k_split_48(shape, res, post, form, monkeypatch):
    """The Cout = 48 forms of the split A-direct kernel: variant 12 (K split over wave pairs, partial accumulators exchanged through LDS) and
    variant 13 (four pixel groups, 16 x 32 tile, one halo buffer).  Selected through EAGLE_CONV_KQ / the tuned table; covered here either way."""
    from eagle_amd import lib
    from oracle import prims as P
    monkeypatch.setenv("EAGLE_CONV_KQ", form)
    n, h, w = shape
    x = _rand((n, h, w, 48), 61)
    wt = _rand((3, 3, 48, 48), 62, (2.0 / (48 * 9)) ** 0.5)
    b = _rand((48,), 63, 0.1)
    r1 = _rand((n, h, w, 48), 64) if res else None
    r2 = _rand((n, h, w, 48), 65) if res == 2 else None
    ref = P.conv2d(x, wt, b, stride=1, pre=0, r1=r1, r2=r2, post=post)
    got = lib.op_conv2d(x, wt, b, 1, 0, r1, r2, post, lib.PREC_F32S)
    err = np.abs(ref - got).max() / max(np.abs(ref).max(), 1e-6)
    assert err < F32S_TOL, f"split K-split conv error {err}"


@pytest.mark.parametrize("cin,cout", [(48, 96), (96, 192), (256, 96), (192, 384), (48, 192), (96, 96), (32, 96)])
@pytest.mark.parametrize("shape", [(3, 37, 45), (2, 5, 70), (1, 16, 16), (2, 135, 240), (1, 1, 1)])
@pytest.mark.parametrize("res,post", [(True, 1), (False, 0), (2, 1)])
def test_conv_split_a_direct_stride2(cin, cout, shape, res, post, monkeypatch):
    """The stride-2 form of the split A-direct kernel (variants 10 / 11: space-to-depth image gathered by the LDS-DMA addressing, 6 K-steps per
    16-channel chunk), forced through EAGLE_CONV_FORCE: odd and even input sizes, partial tiles, a 1 x 1 input, 0 - 2 residual operands."""
    from eagle_amd import lib
    from oracle import prims as P
    if shape[1] * shape[2] > 4000 and (cin > 96 or not res):
        pytest.skip("large map: one representative case")
    monkeypatch.setenv("EAGLE_CONV_FORCE", "16,12,10" if cout % 192 == 0 else "16,6,11")
    n, h, w = shape
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    x = _rand((n, h, w, cin), 71)
    wt = _rand((3, 3, cin, cout), 72, (2.0 / (cin * 9)) ** 0.5)
    b = _rand((cout,), 73, 0.1)
    r1 = _rand((n, ho, wo, cout), 74) if res else None
    r2 = _rand((n, ho, wo, cout), 75) if res == 2 else None
    ref = P.conv2d(x, wt, b, stride=2, pre=0, r1=r1, r2=r2, post=post)
    got = lib.op_conv2d(x, wt, b, 2, 0, r1, r2, post, lib.PREC_F32S)
    err = np.abs(ref - got).max() / max(np.abs(ref).max(), 1e-6)
    assert err < F32S_TOL, f"split stride-2 A-direct conv error {err}"


@pytest.mark.parametrize("cin,cout", [(96, 192), (48, 192), (192, 384), (48, 384), (144, 192), (48, 96), (96, 96), (48, 288)])
@pytest.mark.parametrize("shape", [(3, 37, 45), (2, 5, 70), (1, 16, 16), (2, 68, 120), (1, 1, 1), (2, 130, 129), (1, 9, 200)])
@pytest.mark.parametrize("res,post", [(True, 1), (False, 0), (2, 1)])
def test_conv_split_a_direct_true_stride2(cin, cout, shape, res, post, monkeypatch):
    """Variants 14 / 15: the stride-1 A-direct kernel over an even / odd column-plane halo (true stride 2, no space-to-depth padding; Cin = 48 k;
    Cout = 192 k: four Cout groups; Cout = 96 k: two Cout groups with the K dimension split over wave pairs), forced through EAGLE_CONV_FORCE: odd and even input sizes, partial tiles in both directions, several tiles per row (the 65-column
    halo), a 1 x 1 input, 0 - 2 residual operands."""
    from eagle_amd import lib
    from oracle import prims as P
    if shape[1] * shape[2] > 4000 and (cin > 96 or not res):
        pytest.skip("large map: one representative case")
    monkeypatch.setenv("EAGLE_CONV_FORCE", "16,12,14" if cout % 192 == 0 else "16,6,15")
    n, h, w = shape
    ho, wo = (h + 2 - 3) // 2 + 1, (w + 2 - 3) // 2 + 1
    x = _rand((n, h, w, cin), 81)
    wt = _rand((3, 3, cin, cout), 82, (2.0 / (cin * 9)) ** 0.5)
    b = _rand((cout,), 83, 0.1)
    r1 = _rand((n, ho, wo, cout), 84) if res else None
    r2 = _rand((n, ho, wo, cout), 85) if res == 2 else None
    ref = P.conv2d(x, wt, b, stride=2, pre=0, r1=r1, r2=r2, post=post)
    got = lib.op_conv2d(x, wt, b, 2, 0, r1, r2, post, lib.PREC_F32S)
    err = np.abs(ref - got).max() / max(np.abs(ref).max(), 1e-6)
    assert err < F32S_TOL, f"split true-stride-2 A-direct conv error {err}"


@pytest.mark.parametrize("shape", [(3, 37, 45), (1, 16, 16), (2, 135, 240), (2, 9, 100), (1, 8, 48), (1, 1, 1)])
@pytest.mark.parametrize("res,post", [(True, 1), (False, 0), (2, 1)])
def test_conv_split_8x48_tile(shape, res, post, monkeypatch):
    """Variant 18 of the generic split kernel: six 16-pixel sub-tiles per wave, 8 x 48 output tiles (HRNet's 240-pixel-wide 48-channel branch: 240 = 5 x 48),
    forced through EAGLE_CONV_FORCE: exact and partial tiles in both directions, 0 - 2 residual operands."""
    from eagle_amd import lib
    from oracle import prims as P
    monkeypatch.setenv("EAGLE_CONV_FORCE", "16,3,18")
    n, h, w = shape
    x = _rand((n, h, w, 48), 101)
    wt = _rand((3, 3, 48, 48), 102, (2.0 / (48 * 9)) ** 0.5)
    b = _rand((48,), 103, 0.1)
    r1 = _rand((n, h, w, 48), 104) if res else None
    r2 = _rand((n, h, w, 48), 105) if res == 2 else None
    ref = P.conv2d(x, wt, b, stride=1, pre=0, r1=r1, r2=r2, post=post)
    got = lib.op_conv2d(x, wt, b, 1, 0, r1, r2, post, lib.PREC_F32S)
    err = np.abs(ref - got).max() / max(np.abs(ref).max(), 1e-6)
    assert err < F32S_TOL, f"split 8x48-tile conv error {err}"


def test_conv_split_k_split_exchange_is_reproducible_beside_f16_mfma_waves(state_dicts):
    """VERDICT r3 task 3b.  The K-split A-direct instances (variant 15: every 3x3 stride-2 layer with Cout = 96 k of the split family, e.g.
    HRNet's 48 -> 96 @135x240 -> 68x120) exchange fp32 partial accumulators between the waves of a pair.  Until round 4 that exchange was a
    f32x4 addition = v_pk_add_f32, the instruction class DESIGN.md §8c found unreliable when f16-MFMA-heavy waves share the SIMD; it is scalar
    now (tests/test_isa_guard.py).  Here: the layer, with a residual, 200 times while a second handle runs the fp16 networks on another
    thread — every repeat byte-identical to the first (taken on an idle GPU) and within F32S_TOL of the fp32 oracle."""
    import threading
    from eagle_amd import lib, synth
    from eagle_amd.coordinate_model import CoordinateModel
    from oracle import prims as P
    hs, ys = state_dicts
    x = np.maximum(_rand((2, 135, 240, 48), 41), 0)
    wt = _rand((3, 3, 48, 96), 42, (2.0 / (48 * 9)) ** 0.5)
    b = _rand((96,), 43, 0.1)
    r1 = _rand((2, 68, 120, 96), 44)
    first = lib.op_conv2d(x, wt, b, 2, 0, r1, None, 1, lib.PREC_F32S)            # idle GPU
    ref = P.conv2d(_split_round(x), wt, b, stride=2, pre=0, r1=_split_round(r1), r2=None, post=1)
    assert np.abs(first - ref).max() <= F32S_TOL * np.abs(ref).max()
    co = CoordinateModel(precision="f16", batch=8, hrnet_state_dict=hs, detector_state_dict=ys)
    busy_frames = synth.clip(0, 8)
    stop, batches, err = [False], [0], []

    def busy():
        try:
            while not stop[0]:
                co.process_records(busy_frames)
                batches[0] += 1
        except Exception as e:                       # pragma: no cover
            err.append(repr(e))

    t = threading.Thread(target=busy)
    t.start()
    try:
        bad = 0
        for k in range(200):
            got = lib.op_conv2d(x, wt, b, 2, 0, r1, None, 1, lib.PREC_F32S)
            bad += int(got.tobytes() != first.tobytes())
    finally:
        stop[0] = True
        t.join()
        co.handle.close()
    assert not err, err
    assert batches[0] >= 3, f"the co-runner finished only {batches[0]} batches: no overlap was exercised"
    assert bad == 0, f"{bad} of 200 repeats differ from the idle-GPU result"


def test_conv_split_saturates_instead_of_overflowing():
    """An output beyond the split format's range (|v| > 4094) clips to +-65504 / 16 instead of becoming inf (and NaN one layer later)."""
    from eagle_amd import lib
    x = np.full((1, 8, 8, 16), 100.0, np.float32)
    wt = np.full((3, 3, 16, 16), 1.0, np.float32)
    got = lib.op_conv2d(x, wt, np.zeros(16, np.float32), 1, 0, None, None, 0, lib.PREC_F32S)
    assert np.isfinite(got).all() and got.max() == np.float32(65504.0 / 16.0) and got[0, 0, 0, 0] == np.float32(65504.0 / 16.0)


def test_conv_split_small_and_large_magnitudes():
    """The power-of-two operand scaling keeps the lo parts out of binary16's subnormal range: weights of magnitude 1e-3 and activations of
    magnitude 1e-2 / 1e+2 give the same relative accuracy as O(1) operands."""
    from eagle_amd import lib
    from oracle import prims as P
    for ws, xs in [(1e-3, 1.0), (1.0, 1e-2), (0.05, 1e2), (30.0, 1.0)]:
        x = _rand((1, 20, 37, 48), 41, xs)
        wt = _rand((3, 3, 48, 48), 42, ws)
        b = np.zeros(48, np.float32)
        ref = P.conv2d(x, wt, b, stride=1, pre=0, r1=None, r2=None, post=0)
        got = lib.op_conv2d(x, wt, b, 1, 0, None, None, 0, lib.PREC_F32S)
        err = np.abs(ref - got).max() / np.abs(ref).max()
        assert err < F32S_TOL, f"weights x{ws}, activations x{xs}: error {err}"


@pytest.mark.parametrize("prec", ["f32", "f16", "f32s"])
@pytest.mark.parametrize("shape,upshapes", [((2, 27, 31, 48), [(14, 16), (7, 8), (4, 4)]), ((1, 135, 240, 48), [(68, 120), (34, 60), (17, 30)]),
                                            ((3, 34, 60, 192), [(17, 30)]), ((2, 9, 33, 96), [(5, 17), (1, 1)])])
def test_fuse_sum_parity(prec, shape, upshapes):
    """K4 (one output row per workgroup, gather form) against the oracle in all three families: identical fp32 arithmetic on the stored operands,
    one rounding (binary16 / split pair) at the store; full-size and ragged maps, one to three low-resolution operands, a 1 x 1 operand."""
    from eagle_amd import lib
    from oracle import prims as P
    n, H, W, c = shape
    base = _rand(shape, 1)
    ups = [_rand((n, h, w, c), 2 + i) for i, (h, w) in enumerate(upshapes)]
    q = P.round_f16 if prec == "f16" else _split_round if prec == "f32s" else (lambda a: a)
    y = q(base)
    for u in ups:
        y = y + P.upsample_bilinear_ac(q(u), H, W)
    ref = np.maximum(y, np.float32(0))
    got = lib.op_fuse_sum(base, ups, True, lib.PRECISIONS[prec])
    if prec == "f32":
        assert np.array_equal(ref, got)
    elif prec == "f32s":
        assert np.array_equal(_split_round(ref), got)   # identical fp32 math on the stored (22-bit) operands, one split rounding at the store
    else:
        assert np.array_equal(P.round_f16(ref), got)    # identical fp32 math, one fp16 rounding at the store


@pytest.mark.parametrize("hw", [(720, 1280), (1080, 1920), (360, 640)])
def test_preprocess_parity(hw):
    from eagle_amd import lib, synth
    from oracle import host
    f = np.stack([synth.noise_frame(3, *hw), synth.frame(1, 4, *hw)])
    kp, det = lib.op_preprocess(f, 640, lib.PREC_F32)
    for i in range(2):
        assert np.array_equal(host.preprocess_keypoints(f[i])[0], kp[i])
        assert np.array_equal(host.preprocess_detector(f[i], 640)[0][0], det[i])
    kps, dets = lib.op_preprocess(f, 640, lib.PREC_F32S)      # the split family stores the same fp32 values to 22+ bits
    assert np.array_equal(_split_round(kp), kps) and np.array_equal(_split_round(det), dets)
    assert np.abs(kps - kp).max() <= 2.0 ** -22 * np.abs(kp).max()


def _camera_points(seed, noise=0.0, n_out=0):
    from eagle_amd import synth
    from eagle_amd.pitch import LANDMARKS, on_plane_mask
    rng = np.random.default_rng(seed)
    Hm = synth.camera(seed, 10 * seed)
    m = on_plane_mask()
    world = np.array([[x, y] for (i, _, x, y, z) in LANDMARKS if m[i]], np.float64)
    img = synth.project(Hm, world)
    keep = (img[:, 0] >= 0) & (img[:, 0] < 1280) & (img[:, 1] >= 0) & (img[:, 1] < 720)
    img, world = img[keep], world[keep]
    img = np.floor(img + rng.normal(0, noise, img.shape))
    for k in rng.choice(len(img), size=min(n_out, len(img)), replace=False):
        img[k] = rng.uniform(0, 700, 2)
    return img.astype(np.float32), world.astype(np.float32)


@pytest.mark.parametrize("seed,noise,n_out", [(0, 0.0, 0), (1, 0.7, 0), (2, 0.5, 4), (3, 1.0, 7), (4, 0.0, 2), (5, 2.0, 14), (6, 0.3, 18)])
def test_find_homography_bit_exact(seed, noise, n_out):
    from eagle_amd import lib
    from oracle import prims as P
    img, world = _camera_points(seed, noise, n_out)
    H0, m0 = P.find_homography_ransac(img, world, 5.0)
    H1, m1 = lib.op_find_homography(img, world, 5.0)
    assert (H0 is None) == (H1 is None)
    if H0 is not None:
        assert np.array_equal(m0, m1)
        assert np.array_equal(H0, H1), f"H differs: {np.abs(H0 - H1).max()}"


def test_find_homography_degenerate():
    from eagle_amd import lib
    pts = np.array([[0, 0], [1, 1], [2, 2], [3, 3], [4, 4]], np.float32)
    assert lib.op_find_homography(pts, pts)[0] is None          # collinear: no valid subset
    assert lib.op_find_homography(pts[:3], pts[:3])[0] is None  # fewer than 4 points


@pytest.mark.parametrize("seed,n", [(0, 5), (1, 9), (2, 30), (3, 57), (4, 87)])
def test_find_homography_garbage_points_full_2000_iterations(seed, n):
    """Geometrically meaningless correspondences (what random-weight heat-maps give): RANSAC runs to its iteration cap,
    through several parallel rounds of the precomputed cv::RNG stream; still bit-identical to the sequential oracle."""
    from eagle_amd import lib
    from oracle import prims as P
    rng = np.random.default_rng(100 + seed)
    img = np.floor(rng.uniform(0, 1280, (n, 2))).astype(np.float32)
    world = rng.uniform(0, 105, (n, 2)).astype(np.float32)
    H0, m0 = P.find_homography_ransac(img, world, 5.0)
    H1, m1 = lib.op_find_homography(img, world, 5.0)
    assert (H0 is None) == (H1 is None)
    if H0 is not None:
        assert np.array_equal(m0, m1) and np.array_equal(H0, H1)

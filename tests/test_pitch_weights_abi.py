"""CPU tests: data tables, synthetic weights, and the C-ABI surface (load + symbols + struct sizes; no compute)."""
import ctypes
import json
import os
import re

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)


def test_pitch_tables_match_reference_dump():
    from eagle_amd import pitch
    g = json.load(open(os.path.join(HERE, "golden", "pitch_tables.json")))
    assert {int(k): v for k, v in g["INTERSECTION_TO_PITCH_POINTS"].items()} == pitch.INTERSECTION_TO_PITCH_POINTS
    assert list(pitch.NOT_ON_PLANE) == g["NOT_ON_PLANE"]
    assert [[k, list(v)] for k, v in pitch.GROUND_TRUTH_POINTS.items()] == g["GROUND_TRUTH_POINTS"]   # values AND insertion order
    assert sum(pitch.on_plane_mask()) == 53


def test_generated_device_table_is_current():
    txt = open(os.path.join(ROOT, "eagle_amd", "csrc", "pitch_table.h")).read()
    assert "#define PT_NY 19" in txt and "#define PT_NX 19" in txt
    from eagle_amd import pitch
    for _, _, x, y, _ in pitch.LANDMARKS:
        assert f"{{{x!r}, {y!r}}}" in txt


def test_synthetic_weights_counts_and_determinism():
    from eagle_amd import weights as W
    sd = W.make_hrnet_state_dict(0)
    assert len(sd) == 1754 and W.n_params(sd) == 63_619_593
    assert len(W.hrnet_convs()) == 293
    for v, n in (("n", 3_157_200), ("s", 11_166_560), ("m", 25_902_640), ("l", 43_691_520), ("x", 68_229_648)):
        assert W.param_count(W.yolo_convs(v, 80), 16) == n               # published YOLOv8 parameter counts
    assert W.param_count(W.yolo_convs("n", 5), 16) == 3_011_823
    a, b = W.make_yolo_state_dict("n", 0), W.make_yolo_state_dict("n", 0)
    assert all(np.array_equal(a[k], b[k]) for k in a)
    c = W.make_yolo_state_dict("n", 1)
    assert not np.array_equal(a["model.0.conv.weight"], c["model.0.conv.weight"])


def test_abi_library_loads_and_exports_every_declared_symbol():
    from eagle_amd import lib
    L = lib.load()
    hdr = open(os.path.join(ROOT, "include", "eagle.h")).read()
    declared = set(re.findall(r"^\s*(?:int|void|const char\*)\s+(eagle_\w+)\s*\(", hdr, re.M))
    assert declared and declared == set(lib.EXPORTS), declared ^ set(lib.EXPORTS)
    for s in declared:
        assert hasattr(L, s), s
    cfg_sz, res_sz, det_sz, kp_sz = lib.abi_sizes()
    assert cfg_sz == ctypes.sizeof(lib.EagleConfig)
    assert (res_sz, det_sz, kp_sz) == (lib.RESULT_DTYPE.itemsize, lib.DET_DTYPE.itemsize, lib.KP_DTYPE.itemsize)
    cfg = lib.default_config()
    assert (cfg.frame_h, cfg.frame_w, cfg.det_imgsz) == (720, 1280, 640)
    assert (cfg.keypoint_conf, cfg.detector_conf, cfg.ransac_thresh) == (0.3, 0.35, 5.0)
    assert abs(cfg.detector_floor - 0.15) < 1e-7 and abs(cfg.nms_iou - 0.7) < 1e-7
    assert (cfg.ransac_max_iters, cfg.lm_iters) == (2000, 10)


def test_product_path_fails_loudly_without_gpu_or_library():
    import pytest
    from eagle_amd import lib
    try:
        import torch
        has_gpu = torch.cuda.device_count() > 0
    except Exception:
        has_gpu = False
    if has_gpu:
        pytest.skip("GPU present")
    with pytest.raises(lib.EagleError):
        lib.Handle()


def test_product_does_not_import_oracle():
    for dp, _, fs in os.walk(os.path.join(ROOT, "eagle_amd")):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                src = open(os.path.join(dp, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle", src, re.M), f
                assert "eo_prims" not in src or f.endswith((".hip", ".h")) and "oracle/eo_prims.c" in src, f


def test_default_config_semantics_and_new_fields():
    """Round 4: eagle_default_config = split-precision key-points + exact-fp32 detector; a caller that picks another family without naming the detector's
    gets the detector in that family; explicit choices are kept; the saturation switch defaults to "fail loudly"; the header's error code and the Python
    exception exist; EagleTimings carries the saturation counters inside its fixed size."""
    import ctypes as C
    from eagle_amd import lib
    c = lib.default_config()
    assert (c.precision, c.det_precision, c.allow_saturation) == (lib.PREC_F32S, lib.DET_PREC_AUTO, 0)
    # round 5: the default is resolved INSIDE the C library (eagle_create -> eagle_resolve_config), so a C-ABI caller that takes
    # eagle_default_config and only sets cfg.precision = EAGLE_PREC_F16 gets both networks in the fast family; lib.default_config is a pass-through
    raw = lib.EagleConfig(); lib.load().eagle_default_config(C.byref(raw))
    raw.precision = lib.PREC_F16
    assert raw.det_precision == lib.DET_PREC_AUTO and lib.resolve_config(raw).det_precision == 0
    assert lib.resolve_config(c).det_precision == lib.PREC_F32 + 1
    assert lib.resolve_config(lib.default_config(precision=lib.PREC_F16)).det_precision == 0
    assert lib.resolve_config(lib.default_config(precision=lib.PREC_F32)).det_precision == 0
    assert lib.resolve_config(lib.default_config(precision=lib.PREC_F32S)).det_precision == lib.PREC_F32 + 1
    assert lib.resolve_config(lib.default_config(precision=lib.PREC_F16, det_precision=lib.PREC_F32S + 1)).det_precision == lib.PREC_F32S + 1
    assert lib.resolve_config(lib.default_config(precision=lib.PREC_F32S, det_precision=0)).det_precision == 0
    # round 6: the mixed detector (split trunk, exact last C2f per level + Detect) is an explicit choice, kept as given, and the header's constant is lib.py's
    assert lib.resolve_config(lib.default_config(precision=lib.PREC_F32S, det_precision=lib.DET_PREC_MIXED)).det_precision == lib.DET_PREC_MIXED
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "eagle.h")).read()
    assert re.search(r"#define\s+EAGLE_DET_PREC_MIXED\s+%d\b" % lib.DET_PREC_MIXED, hdr)
    # small-batch mode (round 5): use_graph / multi_stream default to "auto" and resolve from the batch
    assert (c.use_graph, c.multi_stream) == (lib.AUTO, lib.AUTO)
    os.environ.pop("EAGLE_MULTI_STREAM", None)
    for B, want in ((1, (1, 1)), (lib.SMALL_BATCH, (1, 1)), (16, (1, 1)), (lib.SMALL_BATCH + 1, (0, 1)), (50, (0, 1))):      # (round 6: branch streams at every batch)
        r = lib.resolve_config(lib.default_config(batch=B))
        assert (r.use_graph, r.multi_stream) == want, B
    r = lib.resolve_config(lib.default_config(batch=1, use_graph=0, multi_stream=0))
    assert (r.use_graph, r.multi_stream) == (0, 0)
    assert "EAGLE_SMALL_BATCH %d " % lib.SMALL_BATCH in open(os.path.join(ROOT, "include", "eagle.h")).read()
    assert lib.default_config(allow_saturation=1).allow_saturation == 1
    assert issubclass(lib.EagleRangeError, lib.EagleError) and lib.E_RANGE == -8
    hdr = open(os.path.join(ROOT, "include", "eagle.h")).read()
    assert "#define EAGLE_E_RANGE (-8)" in hdr and "allow_saturation" in hdr and "sat_events" in hdr
    assert C.sizeof(lib.EagleTimings) == 4 * 4 + 8 + 8 * 4          # total_ms, conv_ms, 2 counts, conv_flop, (sat_events, sat_frames, reserved[6])
    assert lib.abi_sizes()[0] == C.sizeof(lib.EagleConfig)


def test_torch_after_loading_the_library_is_refused():
    """it is the dlopen of libeagle_hip.so (its NEEDED libamdhip64.so.7 maps /opt/rocm's runtime), not the first handle, after which a
    torch import maps a second runtime: load() / default_config() / comm_unique_id() alone must trip the guard."""
    import subprocess
    import sys
    root = ROOT
    code = ("import sys, numpy as np\n"
            "from eagle_amd import lib, shard\n"
            "lib.default_config()\n"
            "assert 'torch' not in sys.modules\n"
            "try:\n"
            "    shard.gather_records(np.zeros(1, lib.RESULT_DTYPE), 2, 0, 2, transport='dist')\n"
            "except lib.EagleError as e:\n"
            "    assert 'torch must be imported before' in str(e); print('REFUSED')\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and "REFUSED" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])

"""-m gpu: edge cases of the per-frame path (the reference has no tests; these are the cases its loop body guards with
try/except or explicit checks: no detections, < 4 plane points -> no homography, empty input, ragged batches)."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def records_equal(a, b):
    """Field-wise equality (np.concatenate leaves struct padding bytes of structured arrays uninitialised)."""
    def eq(x, y):
        if x.dtype.names:
            return all(eq(x[n], y[n]) for n in x.dtype.names)
        return np.array_equal(x, y)
    return a.shape == b.shape and eq(a, b)


@pytest.fixture(scope="module")
def model(state_dicts):
    from eagle_amd.coordinate_model import CoordinateModel
    hs, ys = state_dicts
    m = CoordinateModel(precision="f16", batch=4, hrnet_state_dict=hs, detector_state_dict=ys)
    yield m
    m.handle.close()


def test_empty_clip_and_ragged_batches(model):
    from eagle_amd import lib, synth
    assert len(model.process_records(np.zeros((0, 720, 1280, 3), np.uint8))) == 0
    frames = np.stack([synth.frame(2, t) for t in range(6)])       # 6 frames through batch 4: 4 + 2
    a = model.process_records(frames)
    b = np.concatenate([model.process_records(frames[i:i + 1]) for i in range(6)])
    assert a.dtype == lib.RESULT_DTYPE and records_equal(a, b)


def test_host_fed_equals_device_fed(model):
    from eagle_amd import synth
    frames = np.stack([synth.frame(5, t) for t in range(5)])
    host = model.handle.process(frames)
    d = model.handle.upload(frames)
    dev = model.handle.process_device(d, len(frames))
    model.handle.free(d)
    assert records_equal(host, dev)


def _strided_view(buf, n, h, w, row_stride, frame_stride, offset=0):
    return np.lib.stride_tricks.as_strided(buf[offset:], shape=(n, h, w, 3), strides=(frame_stride, row_stride, 3, 1))


@pytest.mark.parametrize("memory", ["pinned", "pageable"])
def test_strided_frames_give_the_records_of_the_dense_call(model, memory):
    """VERDICT r4 task 5: the strided half of eagle_process_frames (include/eagle.h: frame_stride / row_stride in bytes) — the frame the reference
    hands over is whatever view the caller holds (cm.py:568).  (i) a pinned buffer with row_stride > 3 w and frame_stride > h row_stride takes
    hipMemcpy2DAsync per frame, (ii) the same layout in pageable memory takes the per-row memcpy into the pinned ring; (iii) a crop out of a wider,
    taller surface.  Five frames through batch 4 (a ragged second batch); the padding bytes hold garbage.  Records byte-identical to the dense call."""
    from eagle_amd import synth
    h, w = 720, 1280
    frames = np.stack([synth.frame(3, t) for t in range(5)])
    dense = model.handle.process(frames)
    rs, off = 3 * w + 64, 192
    fs = h * rs + 4096
    nbytes = off + 5 * fs
    buf = model.handle.host_buffer(nbytes) if memory == "pinned" else np.empty(nbytes, np.uint8)
    try:
        buf[:] = 0xAB
        v = _strided_view(buf, 5, h, w, rs, fs, off)
        v[:] = frames
        assert not v.flags.c_contiguous and v.strides == (fs, rs, 3, 1)
        assert records_equal(model.handle.process(v), dense)
        # a crop of a wider, taller decoder surface: [n, 800, 1400, 3], rows 40..760, columns 60..1340
        if memory == "pageable":
            surf = np.full((5, 800, 1400, 3), 0x5C, np.uint8)
            crop = surf[:, 40:760, 60:1340]
            crop[:] = frames
            assert crop.strides == (800 * 1400 * 3, 1400 * 3, 3, 1)
            assert records_equal(model.handle.process(crop), dense)
            assert records_equal(model.handle.process(crop[2]), dense[2:3])          # one frame of it (ndim 3)
    finally:
        if memory == "pinned":
            model.handle.host_free(buf)


def test_bad_strides_are_rejected(model):
    """A row stride below one row of pixels, a frame stride below one frame, negative strides: EAGLE_E_INVALID, nothing is read."""
    from eagle_amd import lib
    frames = np.zeros((2, 720, 1280, 3), np.uint8)
    for fs, rs in ((0, 3 * 1280 - 1), (720 * 3 * 1280 - 1, 0), (0, -3 * 1280), (-720 * 3 * 1280, 0), (719 * 3 * 1280, 3 * 1280)):
        with pytest.raises(lib.EagleError, match=r"\(-1\)"):
            model.handle.process(frames, strides=(fs, rs))
    model.handle.process(frames, strides=(720 * 3 * 1280, 3 * 1280))             # the dense strides spelled out are fine


def test_wrong_frame_shape_is_rejected(model):
    from eagle_amd import lib
    with pytest.raises(lib.EagleError):
        model.process_records(np.zeros((1, 360, 640, 3), np.uint8))


def test_blank_and_saturated_frames_give_consistent_records(model):
    """Uniform frames: whatever the random-weight networks answer, the record must be internally consistent."""
    frames = np.stack([np.zeros((720, 1280, 3), np.uint8), np.full((720, 1280, 3), 255, np.uint8)])
    for rec in model.process_records(frames):
        n, k = int(rec["n_det"]), int(rec["n_kp"])
        assert 0 <= n <= 300 and 0 <= k <= 87 and n <= int(rec["n_candidates"])
        d = rec["det"][:n]
        assert np.all(np.diff(d["conf"]) <= 0)                                   # NMS order = descending confidence
        assert np.all((d["x1"] >= 0) & (d["x2"] <= 1280) & (d["y1"] >= 0) & (d["y2"] <= 720))
        persons = np.isin(d["cls"], (0, 1))
        assert np.array_equal(d["id"][persons], np.nonzero(persons)[0])          # tracker-less IDs (cm.py:598-616)
        balls = d["cls"] == 2
        assert np.array_equal(d["id"][balls], np.arange(balls.sum()))            # enumerate index (cm.py:619-627)
        assert np.all(d["id"][~persons & ~balls] == -1)
        kp = rec["kp"][:k]
        assert len(set(zip(kp["x"].tolist(), kp["y"].tolist()))) >= len(set(kp["label"].tolist())) - 30 or k == 0
        assert len(set(kp["label"].tolist())) == k                               # one entry per landmark label
        if not rec["H_valid"]:
            assert not rec["bounds_valid"] and not d["in_bounds"].any()
        else:
            assert rec["H"][8] == 1.0 and kp["on_plane"].sum() >= 4 and kp["inlier"].sum() >= 4
            inb = d["in_bounds"].astype(bool)
            assert np.all((d["pitch_x"][inb] >= 0) & (d["pitch_x"][inb] <= 105) & (d["pitch_y"][inb] >= 0) & (d["pitch_y"][inb] <= 68))


def test_high_keypoint_threshold_means_no_homography(state_dicts):
    """keypoint_conf above every sigmoid maximum -> no keypoints -> H_valid = 0, every object keeps its image foot point
    (the `H_use is None` branch, cm.py:379-380)."""
    from eagle_amd import records, synth
    from eagle_amd.coordinate_model import CoordinateModel
    hs, ys = state_dicts
    m = CoordinateModel(keypoint_conf=1.5, precision="f16", batch=1, hrnet_state_dict=hs, detector_state_dict=ys)
    rec = m.process_records(synth.frame(0, 1)[None])[0]
    m.handle.close()
    assert rec["n_kp"] == 0 and not rec["H_valid"]
    ref = records.to_reference_dict(rec)
    assert ref["Boundaries"] == [None] * 4 and ref["Keypoints"] == {}
    for objs in ref["Coordinates"].values():
        for o in objs.values():
            assert o["Transformed_Coordinates"] is None and "Image_Bottom_center" in o


def test_processor_process_api(state_dicts):
    from eagle_amd import synth
    from eagle_amd.processor import Processor
    hs, ys = state_dicts
    p = Processor(precision="f16", batch=1, hrnet_state_dict=hs, detector_state_dict=ys)
    out = p.process(synth.frame(0, 9))
    p.model.handle.close()
    assert set(out) >= {"players", "ball", "H"}
    assert out["H"] is None or out["H"].shape == (3, 3)
    for pid, pl in out["players"].items():
        assert isinstance(pid, int) and len(pl["BBox"]) == 4 and pl["Type"] in ("Player", "Goalkeeper")


def test_cli_writes_reference_schema(tmp_path):
    """configs[0]: 10-frame clip at --fps 5 -> raw_coordinates.json with the schema of docs/data.md:20-41."""
    import json
    from eagle_amd import cli
    assert cli.main(["--frames", "10", "--fps", "5", "--out", str(tmp_path), "--batch", "4", "--synthetic-weights"]) == 0
    d = json.load(open(tmp_path / "raw_coordinates.json"))
    assert sorted(d, key=int) == [str(i) for i in range(10)]
    for i, rec in d.items():
        assert set(rec) == {"Coordinates", "Time", "Keypoints", "Boundaries"}
        assert rec["Time"] == f"{int(i) // 5 // 60:02d}:{int(i) // 5 % 60:02d}"
        assert len(rec["Boundaries"]) == 4
        for cname, objs in rec["Coordinates"].items():
            assert cname in ("Player", "Goalkeeper", "Ball")
            for oid, o in objs.items():
                int(oid)
                assert len(o["BBox"]) == 4 and all(isinstance(v, int) for v in o["BBox"]) and 0 < o["Confidence"] <= 1
                tc = o["Transformed_Coordinates"]
                assert (tc is None and len(o["Image_Bottom_center"]) == 2) or (0 <= tc[0] <= 105 and 0 <= tc[1] <= 68)
        for label, xy in rec["Keypoints"].items():
            assert isinstance(label, str) and len(xy) == 2


def test_reference_cadence_homography_every_5th_frame(state_dicts):
    """main.py:27's exact call at --fps 5: get_coordinates(frames, 5, num_homography=1, num_keypoint_detection=3).
    The GPU result must equal the oracle's restatement of the reference loop (pinned by cadence_golden.json) fed with
    the same per-frame key-points / detections."""
    from eagle_amd import synth
    from eagle_amd.coordinate_model import CoordinateModel
    from eagle_amd.pitch import INTERSECTION_TO_PITCH_POINTS
    from oracle import host, pipeline
    hs, ys = state_dicts
    frames = np.stack([synth.frame(4, 3 * t) for t in range(7)])
    m = CoordinateModel(precision="f16", batch=3, hrnet_state_dict=hs, detector_state_dict=ys)
    got = m.get_coordinates(frames, 5, num_homography=1, num_keypoint_detection=3, verbose=False)
    recs = m.process_records(frames)
    m.handle.close()
    per_frame = []
    for r in recs:
        kps = {INTERSECTION_TO_PITCH_POINTS[int(k["label"])]: (int(k["x"]), int(k["y"])) for k in r["kp"][: int(r["n_kp"])] if not k["synthesized"]}
        n = int(r["n_det"])
        dets = np.stack([r["det"][k][:n].astype(np.float32) for k in ("x1", "y1", "x2", "y2", "conf", "cls")], 1)
        per_frame.append((kps, host.objects_from_detections(dets, 720, 1280)))
    ref = pipeline.loop_records(per_frame, 5, 1, 720, 1280)

    def canon(d):
        if isinstance(d, dict):
            return {str(k): canon(v) for k, v in d.items() if not str(k).startswith("_")}
        if isinstance(d, (list, tuple)):
            return [canon(v) for v in d]
        return float(d) if isinstance(d, (np.floating, float)) else (int(d) if isinstance(d, np.integer) else d)
    assert canon(got) == canon(ref)
    assert len(got) == 7 and all(set(v) == {"Coordinates", "Time", "Keypoints", "Boundaries"} for v in got.values())


def test_cli_at_25_fps_uses_the_flow_cadence(tmp_path):
    """main.py:27's call at 25 fps: HRNet on every 8th frame, optical flow in between, one homography per second; the file must
    carry the reference's value types (flowed key-points are numpy integers there, which json.dump(default=float) writes as floats)."""
    import json
    from eagle_amd import cli
    assert cli.main(["--frames", "11", "--fps", "25", "--out", str(tmp_path), "--batch", "4", "--synthetic-weights"]) == 0
    d = json.load(open(tmp_path / "raw_coordinates.json"))
    assert sorted(d, key=int) == [str(i) for i in range(11)]
    assert all(set(r) == {"Coordinates", "Time", "Keypoints", "Boundaries"} for r in d.values())
    every = cli.main(["--frames", "11", "--fps", "25", "--out", str(tmp_path / "e"), "--batch", "4", "--every-frame", "--synthetic-weights"])
    assert every == 0 and len(json.load(open(tmp_path / "e" / "raw_coordinates.json"))) == 11


def test_cfg3_fp16_family_builds_and_runs():
    """configs[2] in the fp16 family: 1920x1080 frames, YOLOv8-l @960.  The tuned per-layer table must only hand a layer to a
    kernel variant that implements its epilogue (the weight-stationary variant has no SiLU epilogue: YOLO's 64->64 3x3 layers
    share their shape with HRNet's stem convs)."""
    from eagle_amd import synth
    from eagle_amd.coordinate_model import CoordinateModel
    m = CoordinateModel(precision="f16", batch=2, frame_hw=(1080, 1920), detector="l", det_imgsz=960)
    try:
        frames = np.stack([synth.frame(0, 0, 1080, 1920), synth.noise_frame(1, 1080, 1920)])
        recs = m.process_records(frames)
        assert len(recs) == 2 and all(0 <= int(r["n_det"]) <= 300 for r in recs)
        assert all(np.isfinite(r["hm_score"]).all() for r in recs)
    finally:
        m.handle.close()


def test_two_ranks_on_one_gpu_equal_a_single_rank_run():
    """The N > 1 path end to end with REAL records before any multi-GPU node exists: two fresh processes, each with its own handle
    on device 0, frame-sharded (stateless path) and clip-sharded (flow cadence); gathered records == single-rank records."""
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", str(port), os.path.join(root, "tools", "shard_check.py")], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0 and "SHARD_CHECK_OK world=2" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def _torchrun(nproc, script_args, extra_env=None, timeout=1500):
    import socket
    import subprocess
    import sys
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", **(extra_env or {}))
    return subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
                           "--master-port", str(port)] + script_args, capture_output=True, text=True, timeout=timeout, env=env, cwd=root)


def test_eight_ranks_on_one_gpu_ragged_frames_and_nine_clips():
    """8-GPU pre-flight without the node: eight fresh processes share device 0, 25 frames (chunks of 4: rank 6 holds one frame, rank 7 none),
    nine clips for eight ranks in the cadence; the gathered records equal a single-rank run (tools/shard_check.py)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = _torchrun(8, [os.path.join(root, "tools", "shard_check.py")], {"SHARD_CHECK_FRAMES": "25", "SHARD_CHECK_BATCH": "2", "OMP_NUM_THREADS": "4"})
    assert r.returncode == 0 and "SHARD_CHECK_OK world=8 frames=25" in r.stdout, (r.stdout[-2000:], r.stderr[-3000:])


def test_bench_command_path_at_eight_ranks_on_one_gpu():
    """The exact command the driver launches for N = 8 (torch.distributed.run ... bench.py --gpus 8 --steps K --warmup W), with the two
    developer switches that let eight ranks share one device (--shared-gpu, --backend gloo): argument parsing, the barrier-bracketed timed
    region, the gather inside it, max-over-ranks and the one JSON line have all executed once before a node exists."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = _torchrun(8, [os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1", "--batch", "4", "--shared-gpu", "--backend", "gloo",
                      "--no-cpu-baseline"], {"OMP_NUM_THREADS": "4"})
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 8 and res["steps"] == 2 and res["warmup"] == 1 and res["scaling"] == "weak" and res["value"] > 0
    assert res["config"]["frames_total"] == 8 * 2 * 4 and res["config"]["gather"] == "dist"


def test_bench_self_launch_reports_the_gpu_count_it_was_asked_for():
    """VERDICT r5 task 3.  `python bench.py --gpus 2 ...` with NO launcher and no WORLD_SIZE: the process starts the two ranks itself (fresh children through
    torch.distributed.run) and relays rank 0's one line — the same line shape as the torchrun invocation above, `n_gpus` = what was asked for (until round 5
    this command silently measured one GPU and said "n_gpus": 1).  Two ranks share device 0 here (--shared-gpu --backend gloo)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    env["OMP_NUM_THREADS"] = "4"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--batch", "4", "--shared-gpu", "--backend", "gloo",
                        "--no-cpu-baseline"], capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 2 and res["steps"] == 2 and res["warmup"] == 1 and res["scaling"] == "weak" and res["value"] > 0
    assert res["config"]["frames_total"] == 2 * 2 * 4


@pytest.mark.parametrize("gather", ["rccl", "dist"])
def test_bench_multirank_process_composition_with_nccl_group_and_library_rccl(gather):
    """VERDICT r3 task 2.  The process composition the driver runs at N = 8 — torch-ROCm imported, an `nccl` process group, then the HIP library and
    its dlopen'ed RCCL communicator in the SAME process — executed at world 1 in a FRESH process that has not touched the GPU before
    (`--force-multirank-path` takes every branch of the WORLD_SIZE > 1 path: init_process_group("nccl"), shard.init_rccl = ncclGetUniqueId ->
    dist.broadcast on the GPU -> ncclCommInitRank, eagle_gather = ncclAllGather inside the timed region, all_reduce(MAX), barrier, destroy).
    rc must be 0 (round 3 saw an abort at interpreter exit when torch was imported next to the RTLD_GLOBAL RCCL) and the line must be there;
    `dist`: the labelled alternative transport, torch.distributed.all_gather on the nccl group."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = _torchrun(1, [os.path.join(root, "bench.py"), "--gpus", "1", "--force-multirank-path", "--backend", "nccl", "--gather", gather, "--steps", "2", "--warmup", "1",
                      "--batch", "4", "--no-cpu-baseline"], {"OMP_NUM_THREADS": "4"}, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, (r.stdout[-2000:], r.stderr[-2000:])
    # ... and NOTHING else on stdout: torch's bundled RCCL prints a version banner through C stdio that used to land behind the JSON line (round 5)
    assert [ln for ln in r.stdout.splitlines() if ln.strip()] == lines, r.stdout[-2000:]
    res = json.loads(lines[0])
    assert res["n_gpus"] == 1 and res["steps"] == 2 and res["value"] > 0 and res["config"]["gather"] == gather, res["config"]
    assert res["config"]["runtime"].startswith("torch-bundled") and set(res["config"]["timed_region_parts_rank0"]) == {"process_ms", "gather_ms", "barrier_ms"}
    assert "bootstrap failed" not in r.stderr, r.stderr[-3000:]


def test_bench_multirank_path_replays_its_step_and_captures_outside_the_timed_region():
    """The multi-rank path creates its handle with use_graph = 2 (replay inside calls of >= 3 steps: the PyTorch wheel's runtime launches slower).  With fewer
    warm-up steps than the three that switch the replay on, the untimed call is widened so that the ~80-ms captures do not land in the timed region."""
    import json
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = _torchrun(1, [os.path.join(root, "bench.py"), "--gpus", "1", "--force-multirank-path", "--backend", "nccl", "--gather", "rccl", "--steps", "3", "--warmup", "1",
                      "--batch", "4", "--no-cpu-baseline"], {"OMP_NUM_THREADS": "4"}, timeout=900)
    assert r.returncode == 0, (r.returncode, r.stdout[-2000:], r.stderr[-4000:])
    res = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][0])
    assert res["warmup"] == 1 and res["steps"] == 3 and res["config"]["hip_graph"] is True and res["config"]["graph_priming_steps"] == 3, res["config"]


def test_default_handle_parity_on_the_runtime_the_multirank_bench_uses():
    """VERDICT r4 task 1a.  With torch imported first (bench.py's WORLD_SIZE > 1 order) the library runs on the HIP / HSA / RCCL copies bundled in the
    PyTorch wheel, not on /opt/rocm's — a different runtime than every other GPU test uses.  A FRESH process does exactly that, proves from
    /proc/self/maps that one copy of each library is mapped and that it is torch's, and runs the exception-free default-handle comparison on the five
    cfg-2 frames against the fp32 CPU oracle, plus eagle_gather on that runtime's RCCL (tests/torch_first_parity.py)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "torch_first_parity.py")], capture_output=True, text=True, timeout=1500, cwd=root,
                       env=dict(os.environ, OMP_NUM_THREADS="8"))
    assert r.returncode == 0 and "TORCH_FIRST_PARITY_OK" in r.stdout, (r.stdout[-2000:], r.stderr[-4000:])
    print(r.stdout[-800:])


def test_bench_default_line_carries_the_measurement_contract():
    """`python bench.py` at N = 1 (small sizes): ONE JSON line with the contract's fields — value from host frames, `roofline` with frac = achieved / peak and
    the dominant kernel named, `cpu_baseline` from the child process, `parity_counters.default` all zero, `saturation` 0 / 0 — and the process never imported
    torch (the child did)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--batch", "4", "--distinct", "4", "--cpu-frames", "1",
                        "--exact-frames", "8", "--fast-frames", "8", "--cfg3-frames", "0"], capture_output=True, text=True, timeout=900, cwd=root,
                       env=dict(os.environ, OMP_NUM_THREADS="8"))
    assert r.returncode == 0, (r.stdout[-1000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["config"]["input"].startswith("pageable host memory") and d["config"]["detector_precision"] == "f32" and d["config"]["keypoint_precision"] == "f32s"
    assert abs(d["value"] - d["host_sources"]["pageable"]) < 1e-6 and d["resident"]["value"] > 0
    rf = d["roofline"]
    assert rf["bound"] == "mfma" and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3 and rf["dominant_kernel"]["avg_us"] > 0 and 0 < rf["dominant_kernel"]["frac"] < 1
    assert d["cpu_baseline"]["kind"] == "port" and d["cpu_baseline"]["value"] > 0 and d["cpu_baseline"]["cores"] >= 1
    pc = d["parity_counters"][d["parity_counters"]["default"]]
    assert all(pc[k] == 0 for k in ("hm_idx", "n_kp", "kp_pixels", "n_det", "det_cls", "det_int_box", "det_pitch_int", "det_id", "H_valid", "det_unmatched")) and pc["dets_compared"] > 0
    assert d["saturation"]["sat_events"] == 0 and d["saturation"]["sat_frames"] == 0
    assert d["detector_convs"]["family"] == "f32" and d["exact_family"]["value"] > 0 and d["fast_family"]["value"] > 0 and d["split_detector"]["value"] > 0


def test_torch_after_the_first_handle_is_refused():
    """eagle_amd.lib.require_torch_first: the code paths of the package that import torch refuse to do so once a handle exists in a process
    that has not imported torch yet (the order that maps two ROCm runtimes)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys, numpy as np\n"
            "from eagle_amd import lib, shard\n"
            "h = lib.Handle(batch=1)\n"
            "assert 'torch' not in sys.modules\n"
            "try:\n"
            "    shard.gather_records(np.zeros(1, lib.RESULT_DTYPE), 2, 0, 2, transport='dist')\n"
            "except lib.EagleError as e:\n"
            "    assert 'torch must be imported before' in str(e); print('REFUSED')\n"
            "h.close()\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and "REFUSED" in r.stdout, (r.stdout[-1000:], r.stderr[-2000:])


def test_three_handles_on_three_threads_capture_their_graphs_concurrently(state_dicts):
    """Round 5: small batches replay their step as a hipGraph, and a capture is exclusive of every other C-ABI call of the process (HIP refuses legacy-stream
    operations anywhere while any stream captures).  Three host threads each create a handle, upload weights and run ragged calls at the same time — creation,
    finalize (legacy-stream copies) and captures of different handles interleave — and every thread's records equal the single-threaded ones."""
    import threading
    from eagle_amd import synth
    from eagle_amd.coordinate_model import CoordinateModel
    hs, ys = state_dicts
    frames = np.stack([synth.frame(1, t) for t in range(5)])
    ref_m = CoordinateModel(precision="f16", batch=2, hrnet_state_dict=hs, detector_state_dict=ys)
    ref = ref_m.process_records(frames)
    ref_m.handle.close()
    out, err = [None] * 3, []

    def work(k):
        try:
            m = CoordinateModel(precision="f16", batch=2, hrnet_state_dict=hs, detector_state_dict=ys)
            a = m.process_records(frames)                    # 2 + 2 + 1: three captures on this handle
            b = m.process_records(frames[k:k + 2])
            m.handle.close()
            out[k] = (a, b)
        except Exception as e:                               # pragma: no cover
            err.append(repr(e))

    ts = [threading.Thread(target=work, args=(k,)) for k in range(3)]
    [t.start() for t in ts]
    [t.join(timeout=600) for t in ts]
    assert not err and all(not t.is_alive() for t in ts), err
    for k in range(3):
        assert records_equal(out[k][0], ref) and records_equal(out[k][1], ref[k:k + 2]), k


def test_graph_instances_are_cached_per_frame_count_and_survive_a_moving_source(state_dicts):
    """round 5 kept ONE graph per pipeline slot, keyed on (source pointer, frame count) — a call whose last step is ragged re-captured twice per
    call (~80 ms each, process-exclusive), and a caller walking a resident clip re-captured on every call.  Now: one instance per (slot, frame count), device-fed
    calls staged into the slot's own buffer.  EagleTimings::graph_captures is cumulative per handle: the second identical call, a call with the same step shapes at
    another source address and a host-fed call must not add a capture; records stay equal to a handle that replays nothing."""
    from eagle_amd import synth
    from eagle_amd.coordinate_model import CoordinateModel
    hs, ys = state_dicts
    frames = np.stack([synth.frame(1, t) for t in range(7)])
    plain = CoordinateModel(precision="f16", batch=3, hrnet_state_dict=hs, detector_state_dict=ys, use_graph=0)
    ref = plain.process_records(frames)
    plain.handle.close()
    m = CoordinateModel(precision="f16", batch=3, hrnet_state_dict=hs, detector_state_dict=ys, use_graph=1)
    h = m.handle
    a = m.process_records(frames)                            # 3 + 3 + 1 frames: slot 0 sees 3 and 1, slot 1 sees 3 -> three instances
    c0 = h.timings().graph_captures
    assert c0 == 3, c0
    b = m.process_records(frames)
    assert h.timings().graph_captures == c0, "an identical second call re-captured"
    d = h.upload(np.concatenate([frames, frames]))           # a resident clip walked at two offsets: another source pointer per call
    out1 = np.zeros(7, a.dtype); out2 = np.zeros(7, a.dtype)
    h.process_device(d, 7, out1)
    fsz = frames[0].nbytes
    import ctypes
    h.process_device(ctypes.c_void_p(d.value + 7 * fsz), 7, out2)
    h.free(d)
    assert h.timings().graph_captures == c0, "a device-fed call at another address re-captured"
    assert h.timings().graph_skipped == 0
    h.close()
    for got in (a, b, out1, out2):
        assert records_equal(got, ref)


def test_a_capture_never_waits_for_a_long_call_of_another_handle(state_dicts):
    """a capture needs the process-wide lock exclusively; round 5 waited for it without a bound, i.e. for every in-flight call of every handle (a
    multi-second clip call; a collective blocked on this very thread).  Now it tries for 10 ms and otherwise runs the step as plain launches.  Thread A keeps a handle
    busy with long calls; thread B creates a handle that wants to capture and must finish its short calls promptly, with records equal to the reference."""
    import threading
    import time
    from eagle_amd import synth
    from eagle_amd.coordinate_model import CoordinateModel
    hs, ys = state_dicts
    long_clip = np.stack([synth.frame(2, t % 5) for t in range(200)])
    short = np.stack([synth.frame(1, t) for t in range(2)])
    ref_m = CoordinateModel(precision="f16", batch=2, hrnet_state_dict=hs, detector_state_dict=ys, use_graph=0)
    ref = ref_m.process_records(short)
    ref_m.handle.close()
    busy = CoordinateModel(precision="f16", batch=25, hrnet_state_dict=hs, detector_state_dict=ys, use_graph=0)
    busy.process_records(long_clip[:25])                     # warm
    stop, err, spans = threading.Event(), [], []

    def hog():
        try:
            while not stop.is_set():
                busy.process_records(long_clip)              # ~0.1 - 0.2 s per call on the fp16 family, back to back
        except Exception as e:                               # pragma: no cover
            err.append(repr(e))

    t = threading.Thread(target=hog)
    t.start()
    try:
        time.sleep(0.05)
        m = CoordinateModel(precision="f16", batch=2, hrnet_state_dict=hs, detector_state_dict=ys, use_graph=1)
        for _ in range(12):
            t0 = time.perf_counter()
            got = m.process_records(short)
            spans.append(time.perf_counter() - t0)
            assert records_equal(got, ref)
        tm = m.handle.timings()
        m.handle.close()
    finally:
        stop.set()
        t.join(timeout=120)
    busy.handle.close()
    assert not err and not t.is_alive(), err
    # every call returned: nothing waited for the other handle's stream of long calls without a bound (a capture either got its turn between two of them or was skipped)
    assert max(spans) < 5.0, spans
    assert tm.graph_captures + tm.graph_skipped >= 1, (tm.graph_captures, tm.graph_skipped)
    print("captures", tm.graph_captures, "skipped", tm.graph_skipped, "slowest call %.3f s" % max(spans))


def test_geometry_bit_identical_while_another_handle_runs_the_fp16_networks():
    """VERDICT r2 task 9: the concurrent-handle check of the LK kernel, extended to the geometry kernel (`post_kernel`: threshold / dedup / line
    synthesis / RANSAC / DLT / LM / projection, all fp64 and bit-identical to the oracle when run alone).  `eagle_op_find_homography` — the same
    device code — on ten cameras in a loop WHILE a second handle keeps the GPU busy with the fp16 (f16-MFMA) networks on another thread: every H
    and every inlier mask must equal the solo result bit for bit, and the busy handle's own records must not change either."""
    import threading
    from eagle_amd import lib, synth, weights
    from eagle_amd.coordinate_model import CoordinateModel
    from eagle_amd.pitch import LANDMARKS, on_plane_mask
    hs, ys = weights.make_hrnet_state_dict(0), weights.make_yolo_state_dict("n", 0)

    def camera_points(seed, noise, n_out):
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

    cases = [camera_points(s, nz, no) for s, nz, no in [(0, 0.0, 0), (1, 0.7, 0), (2, 0.5, 4), (3, 1.0, 7), (4, 0.0, 2), (5, 2.0, 14), (6, 0.3, 18), (7, 1.5, 3), (8, 0.2, 9), (9, 0.9, 1)]]
    solo = [lib.op_find_homography(i, w, 5.0) for i, w in cases]
    co = CoordinateModel(precision="f16", batch=8, hrnet_state_dict=hs, detector_state_dict=ys)
    busy_frames = synth.clip(0, 8)
    stop, batches, err = [False], [0], []

    def busy():
        try:
            ref = co.process_records(busy_frames)
            while not stop[0]:
                r = co.process_records(busy_frames)
                batches[0] += 1
                if any(r[f].tobytes() != ref[f].tobytes() for f in r.dtype.names):
                    err.append("the co-running fp16 path changed its own records")
        except Exception as e:                       # pragma: no cover
            err.append(repr(e))

    t = threading.Thread(target=busy)
    t.start()
    try:
        rounds = 0
        while batches[0] < 10 and not err and rounds < 2000:
            for (i, w), (H0, m0) in zip(cases, solo):
                H1, m1 = lib.op_find_homography(i, w, 5.0)
                assert (H0 is None) == (H1 is None)
                if H0 is not None:
                    assert np.array_equal(H0, H1) and np.array_equal(m0, m1), "geometry kernel result changed next to the fp16 networks"
            rounds += 1
    finally:
        stop[0] = True
        t.join()
        co.handle.close()
    assert not err, err
    assert batches[0] >= 10 and rounds >= 1

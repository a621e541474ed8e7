#!/usr/bin/env python3
"""bench.py — frames/sec of the per-frame path (detect + keypoints + homography) at 1280x720.

One "step" = one pass of the hot path over one device batch of ``--batch`` synthetic frames.  Default run = BASELINE.json configs[1]: a
1000-frame 1280x720 synthetic clip, YOLOv8-n detector + HRNet-W48 keypoint model, one MI355X (20 steps x 50 frames).  The K timed steps are
issued as ONE library call over the K*batch-frame clip so that the library's two-deep pipeline (upload of step i+1 and geometry + record copy of
step i-1 under the networks of step i) is part of what is measured.

``value`` (SURVEY §8d's metric: H2D of the frame -> ... -> record on the host): the clip starts in PAGEABLE HOST memory (numpy) and goes through
``eagle_process_frames`` — worker threads copy each batch into the library's pinned ring, the DMA runs on a copy stream under the previous batch's
networks.  ``resident`` is the same path with the clip already in HBM (``eagle_process_device_frames``; the two differ by < 1 %), ``host_sources``
adds the pinned-source rate.

Default handle = ``eagle_default_config``: key-point network in EAGLE_PREC_F32S ("f32s": every tensor value a (hi, lo) binary16 pair, three fp16 MFMAs
per product — fp32-grade: heat-map maxima / key-points / H equal the fp32 oracle's on every test frame, floats to a few 1e-6), detector in the EXACT
fp32 family (boxes, confidences, classes, NMS order and detection-index ids bit-identical to the fp32 oracle: tests/test_gpu_pipeline.py::
test_default_handle_dense_detector_is_exception_free_*).  ``parity_counters`` counts integer-field differences against the exact family's records on
the distinct frames of the clip (GPU against GPU; the exact family is the one that equals the oracle bit for bit).  Also in the line:
``fast_family`` (fp16), ``exact_family`` (fp32), ``split_detector`` (both networks in f32s: round 3's headline), ``cfg3`` (BASELINE configs[2]:
1000 frames 1920x1080, yolov8l@960), ``roofline`` (MFMA, the convolution family of the key-point precision, with ``dominant_kernel``),
``roofline_conv_layers``, ``roofline_hbm``, ``saturation`` (clipped f32s stores: 0 in a healthy run), ``cpu_baseline``.

Multi-GPU (driver launches one rank per GPU through torch.distributed.run): frames shard by contiguous chunk, weights replicated, no data-path
collective; ONE all-gather of the fixed-size records at the end (inside the timed region); value = total frames of all ranks / max-over-ranks time
("scaling": "weak").  torch is imported only there, and BEFORE the HIP library is loaded (one ROCm runtime per process: eagle_amd/lib.py::
require_torch_first); at N = 1 the process that touches the GPU never imports torch — the CPU baseline runs first, in a child process.

Prints ONE JSON line on rank 0.  The CPU oracle appears here only as the timed ``cpu_baseline`` leg."""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CPU_BASELINE_THREADS = 16          # threads of the torch-CPU restatement timed as cpu_baseline (the best of a one-off sweep on the 256-CPU GPU host; --cpu-threads overrides)
MFMA_PEAK_TFLOPS = 2500.0         # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md
# peak of the convolution family per precision, in ALGORITHMIC TFLOP/s (2 x MAC of the convolution): the split family spends three fp16 MFMA
# products per algorithmic product (hi*hi + hi*lo + lo*hi), so its roof is a third of the dense fp16 MFMA peak
PEAK = {"f16": MFMA_PEAK_TFLOPS, "f32": 157.3, "f32s": MFMA_PEAK_TFLOPS / 3.0}
CONV_KERNEL = {"f16": "conv_f16_ad_kernel / conv_f16_ws_kernel / conv_f16_kernel", "f32": "conv_f32_kernel",
               "f32s": "conv_f16_kernel<..., SPLIT> / conv_split_ad32_kernel / bneck_split_kernel family (3 fp16 MFMA products per algorithmic product)"}


def log(msg):
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def usable_cpus():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def cpu_baseline(hs, ys, frames, n_frames, threads, budget_s=25.0, variant="n", imgsz=640):
    """The oracle's torch-CPU fp32 restatement of S(frame), timed on this host on a bounded sample
    (at most n_frames frames or ~budget_s seconds of CPU work, whichever comes first)."""
    import torch
    torch.set_num_threads(threads)
    os.environ["OMP_NUM_THREADS"] = str(threads)
    from oracle import pipeline
    m = pipeline.OracleModel(hs, ys, variant=variant, imgsz=imgsz, backend="torch")
    t0 = time.perf_counter()
    m.step(frames[0])                                  # warm-up (weight folding, MKLDNN primitives)
    log(f"cpu_baseline warm-up frame {time.perf_counter() - t0:.1f} s ({threads} threads)")
    t0 = time.perf_counter()
    done = 0
    for i in range(n_frames):
        m.step(frames[i % len(frames)], i)
        done += 1
        if time.perf_counter() - t0 > budget_s:
            break
    dt = time.perf_counter() - t0
    return {"value": round(done / dt, 4), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{done} frames of the same synthetic clip through oracle/pipeline.py (torch-CPU fp32 nets + "
                      f"numpy/C host logic), {threads} threads ({usable_cpus()} usable of {os.cpu_count()} logical CPUs), {dt:.1f} s"}


def cpu_baseline_child(a):
    """--cpu-baseline-only: the child process of the default run (torch lives here, never in the process that touches the GPU)."""
    from eagle_amd import synth, weights
    hs = weights.make_hrnet_state_dict(0)
    ys = weights.make_yolo_state_dict(a.detector, 0)
    frames = synth.clip(seed=0, n=min(a.distinct, a.batch), h=a.height, w=a.width)
    threads = a.cpu_threads if a.cpu_threads > 0 else max(1, min(CPU_BASELINE_THREADS, usable_cpus()))
    print(json.dumps(cpu_baseline(hs, ys, frames, a.cpu_frames, threads, variant=a.detector, imgsz=a.imgsz)), flush=True)
    # (all 256 hardware threads of the GPU box were tried once: torch-CPU convolutions collapse to 0.004 frames/s, 247 s for one
    #  frame, profiles/r02b_bench_default_1gpu.json, so the bounded sample stays at 16 threads and says so)


def cpu_baseline_subprocess(a):
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", "--detector", a.detector, "--imgsz", str(a.imgsz), "--height", str(a.height),
           "--width", str(a.width), "--cpu-frames", str(a.cpu_frames), "--distinct", str(a.distinct), "--batch", str(a.batch), "--cpu-threads", str(a.cpu_threads)]
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
        sys.stderr.write(r.stderr)
        return json.loads(r.stdout.strip().splitlines()[-1])
    except Exception as e:                                # reported, never silent
        return {"value": None, "unit": "frames/s", "cores": 0, "kind": "port", "sample": f"cpu baseline child failed: {e!r}"}


def latency_table(lib, weights, hs, ys, a, dev_index, frames, calls, batches=(1, 2, 4, 8), modes=None):
    """Small-batch behaviour of the DEFAULT handle (north_star's API is Processor.process(frame); the reference's caller is a frame-by-frame loop,
    cm.py:277): one `eagle_process_frames` call of B frames from pageable host memory -> records on the host, per call.  For every B a fresh handle
    with EagleConfig.batch = B; median and p99 wall time over `calls` calls (after 10 warm-up calls), frames/s = B / median.  modes: list of
    (label, env overrides, config overrides)."""
    rows = []
    modes = modes or [("default", {}, {})]
    for B in batches:
        for label, env, kw in modes:
            saved = {k: os.environ.get(k) for k in env}
            for k, v in env.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
            try:
                h = lib.Handle(device=dev_index, frame_h=a.height, frame_w=a.width, det_variant=a.detector, det_imgsz=a.imgsz, batch=B,
                               precision=lib.PRECISIONS[a.precision], **kw)
            finally:
                for k, v in saved.items():
                    if v is None:
                        os.environ.pop(k, None)
                    else:
                        os.environ[k] = v
            weights.load_into(h, [hs, ys])
            out = np.zeros(B, lib.RESULT_DTYPE)
            nf = len(frames) - B + 1
            for k in range(10):
                h.process(frames[(k * B) % nf:][:B], out)
            ts = np.empty(calls)
            for k in range(calls):
                f = frames[(k * B) % nf:][:B]
                t0 = time.perf_counter()
                h.process(f, out)
                ts[k] = time.perf_counter() - t0
            h.close()
            ts.sort()
            med, p99 = float(np.median(ts)), float(ts[min(calls - 1, int(np.ceil(0.99 * calls)) - 1)])
            rows.append({"frames_per_call": B, "mode": label, "median_ms": round(med * 1e3, 3), "p99_ms": round(p99 * 1e3, 3), "min_ms": round(float(ts[0]) * 1e3, 3),
                         "frames_per_s": round(B / med, 1), "calls": calls})
            log(f"latency B={B} {label}: median {med * 1e3:.2f} ms, p99 {p99 * 1e3:.2f} ms, {B / med:.1f} frames/s")
    return rows


def self_launch(n, argv, print_only=False):
    """`python bench.py --gpus N` without a launcher: start the N ranks as fresh child processes — `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    --master-addr 127.0.0.1 --master-port P bench.py <the same arguments>`, the command the driver uses — relay rank 0's ONE JSON line to stdout, everything else to
    stderr, and return the launcher's exit code.  Called before any GPU call and without importing the HIP library or torch in this process (a process that has
    initialised the GPU must not be replaced, and need not be: the parent only waits)."""
    import socket
    with socket.socket() as sk:                            # a free rendezvous port on the loop-back interface
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + [x for x in argv if x != "--print-launch"]
    if print_only:
        print(json.dumps({"launch": cmd}), flush=True)
        return 0
    log(f"--gpus {n} without a launcher: starting {n} ranks: {' '.join(cmd)}")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=None, text=True, env=env)
    line = None
    for ln in p.stdout:                                    # rank 0 prints exactly one JSON line; anything else a native library wrote to stdout goes to stderr
        t = ln.strip()
        ok = False
        if t.startswith("{") and t.endswith("}"):
            try:
                ok = "metric" in json.loads(t)
            except ValueError:
                ok = False
        if ok:
            line = t
        else:
            sys.stderr.write(ln)
    rc = p.wait()
    if rc == 0 and line is None:
        log("the ranks exited 0 but printed no result line")
        rc = 1
    if line is not None and rc == 0:
        got = json.loads(line).get("n_gpus")
        if got != n:                                       # cannot happen with the checks in main(); kept loud
            log(f"result line reports n_gpus = {got}, asked for {n}")
            rc = 1
    if line is not None:
        print(line, flush=True)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=50, help="frames per device step")
    ap.add_argument("--detector", default="n")
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--height", type=int, default=720)
    ap.add_argument("--width", type=int, default=1280)
    ap.add_argument("--precision", default="f32s", choices=["f16", "f32", "f32s"],
                    help="family of the key-point network. f32s (default): split precision, fp32-grade; f16: the fast family; f32: the bit-exact family")
    ap.add_argument("--det-precision", default="default", choices=["default", "f16", "f32", "f32s"],
                    help="family of the detector. default: the library's (exact fp32 next to f32s key-points, otherwise the key-point family)")
    ap.add_argument("--distinct", type=int, default=200, help="distinct synthetic frames generated per rank (tiled to the clip): 200 since round 5 — the data-dependent stages "
                    "(NMS candidate counts, RANSAC iteration counts, key-point dedup) see 200 different frames, and parity_counters compares all of them with the exact family")
    ap.add_argument("--cpu-frames", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-only", action="store_true", help="(internal) time the CPU restatement and print its JSON object")
    ap.add_argument("--graph", action="store_true", help="replay the network phase of every step as a hipGraph (the multi-rank path does it by default inside calls of >= 3 steps)")
    ap.add_argument("--no-graph", action="store_true", help="plain launches only")
    ap.add_argument("--all-layers", action="store_true", help="roofline_conv_layers lists every convolution layer shape instead of the ten heaviest")
    ap.add_argument("--no-extras", action="store_true", help="skip resident / host_sources / families / cfg3 / roofline_hbm (profiling runs)")
    ap.add_argument("--exact-frames", type=int, default=1000, help="frames of the fp32 exact-family run (0: skip)")
    ap.add_argument("--fast-frames", type=int, default=1000, help="frames of the fp16 fast-family run (0: skip)")
    ap.add_argument("--cfg3-frames", type=int, default=1000, help="frames of the configs[2] run (1920x1080, yolov8l@960; 0: skip)")
    ap.add_argument("--realistic-frames", type=int, default=1000, help="frames of the realistic-workload run (peaked key-point head + sparse detector; 0: skip)")
    ap.add_argument("--cpu-threads", type=int, default=0, help="threads of the cpu_baseline leg (0: the default found by the sweep in profiles/r06_cpu_baseline_thread_sweep.txt)")
    ap.add_argument("--cadence", type=int, default=0, metavar="FPS", help="also time the reference's default cadence on the same clip: get_coordinates(frames, FPS, num_homography=1, "
                    "num_keypoint_detection=3) = HRNet every int(FPS/3)-th frame, optical-flow propagation in between (stateful; reported as reference_cadence, never as value)")
    ap.add_argument("--latency-calls", type=int, default=200, help="calls per row of the small-batch latency table (0: skip)")
    ap.add_argument("--latency-only", action="store_true", help="developer: print only the small-batch latency table (all modes: multi-stream / hipGraph on and off)")
    ap.add_argument("--print-launch", action="store_true", help="with --gpus N > 1 and no launcher: print the command the parent would start (one JSON object) and exit")
    ap.add_argument("--gather", default="rccl", choices=["rccl", "dist"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend (gloo + --shared-gpu: dev test of the multi-rank path on one GPU)")
    ap.add_argument("--shared-gpu", action="store_true", help="every rank uses HIP device 0 (developer test only)")
    ap.add_argument("--force-multirank-path", action="store_true",
                    help="take the WORLD_SIZE > 1 code path at world 1 (process group, library RCCL bootstrap, gather inside the timed region, max over ranks): "
                         "the one-GPU regression test of the process composition the driver runs at N = 8")
    a = ap.parse_args()
    import faulthandler
    faulthandler.enable()                                  # a native crash (HIP / RCCL / the library) leaves a Python-level trace on stderr instead of nothing
    # ONE JSON line on stdout, whatever the native libraries print: torch's bundled RCCL writes a five-line version banner to C stdout (block-buffered when
    # stdout is a pipe, so it comes out at exit, BEHIND the JSON line — round 5's first multirank run was mis-parsed that way).  fd 1 is pointed at stderr
    # for the life of the process; the result line goes to a private duplicate of the original stdout.
    if a.cpu_baseline_only:                                # (the child of the default run: prints its own one-line JSON object for the parent)
        cpu_baseline_child(a)
        return
    # --gpus N against the launcher's environment, BEFORE anything touches the GPU or loads the HIP library (VERDICT r5 task 3):
    #   WORLD_SIZE unset, N > 1 : this process is not a rank — it becomes the launcher (self_launch: fresh child processes through torch.distributed.run, never exec)
    #   WORLD_SIZE set, != N    : a mismatch is an error, never a silent one-GPU measurement labelled n_gpus = 1
    if "WORLD_SIZE" not in os.environ:
        if a.gpus > 1:
            sys.exit(self_launch(a.gpus, sys.argv[1:], print_only=a.print_launch))
    elif int(os.environ["WORLD_SIZE"]) != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but the launcher set WORLD_SIZE={os.environ['WORLD_SIZE']}: refusing to measure a different GPU count than the one asked for")
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    multi = world > 1 or a.force_multirank_path

    cpu_base = None
    if not multi and not a.no_cpu_baseline:                # rank 0 at N = 1 only (the measurement contract); before the GPU is touched, in a child
        cpu_base = cpu_baseline_subprocess(a)
        log(f"cpu_baseline: {cpu_base}")

    dist = torch = None
    dev_index = 0 if a.shared_gpu else local_rank
    # CPU placement BEFORE the first GPU call (and after the CPU baseline child): this rank, the library's copy workers and the HIP runtime's helper
    # threads run on the CPUs of the GPU's NUMA node (sysfs only; eagle_amd/shard.py::bind_rank_cpus, EAGLE_BIND_CPUS=0 disables)
    from eagle_amd import shard as _shard                 # (pure Python: neither torch nor the HIP library is loaded by this import)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    cpu_binding = _shard.bind_rank_cpus(local_rank, local_world, rank_devs=[0] * local_world if a.shared_gpu else None)
    log(f"rank {rank}: cpu binding {cpu_binding}")
    tdev = "cuda" if a.backend == "nccl" else "cpu"
    if multi:
        import torch                                       # BEFORE the HIP library is loaded: one ROCm runtime per process (lib.require_torch_first)
        import torch.distributed as dist
        if "MASTER_ADDR" not in os.environ:                # --force-multirank-path without a launcher
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", "29533"), RANK="0", WORLD_SIZE="1")
        if a.backend == "nccl":
            torch.cuda.set_device(dev_index)
            dist.init_process_group("nccl", device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group("gloo")
            if a.gather == "rccl":
                a.gather = "dist"

    from eagle_amd import lib, shard, synth, weights
    B, K, W = a.batch, a.steps, a.warmup
    hs = weights.make_hrnet_state_dict(0)
    ys = weights.make_yolo_state_dict(a.detector, 0)
    if a.latency_only:
        fr = synth.clip(seed=0, n=max(a.distinct, 16), h=a.height, w=a.width)
        modes = [("1stream", {}, {"multi_stream": 0, "use_graph": 0}), ("multi_stream", {}, {"multi_stream": 1, "use_graph": 0}),
                 ("1stream+graph", {}, {"multi_stream": 0, "use_graph": 1}), ("multi_stream+graph", {}, {"multi_stream": 1, "use_graph": 1})]
        bs = tuple(int(x) for x in os.environ.get("LATENCY_BATCHES", "1,2,4,8").split(","))
        os.write(result_fd, (json.dumps({"latency": latency_table(lib, weights, hs, ys, a, dev_index, fr, a.latency_calls, batches=bs, modes=modes)}) + "\n").encode())
        return
    det_kw = {} if a.det_precision == "default" else {"det_precision": lib.PRECISIONS[a.det_precision] + 1}
    h = lib.Handle(device=dev_index, frame_h=a.height, frame_w=a.width, det_variant=a.detector, det_imgsz=a.imgsz,
                   batch=B, precision=lib.PRECISIONS[a.precision], use_graph=1 if a.graph else 0 if a.no_graph else (2 if multi else lib.AUTO), **det_kw)
    # (multi-rank path: use_graph = 2, the step replayed as a captured hipGraph inside calls of >= 3 steps — BASELINE.json configs[4] names it, and on the PyTorch wheel's ROCm 7.0.2
    #  runtime, which this composition runs on, plain launches cost 2.3 % of the frame rate that the replay gives back: same box 739 -> 751 frames/s against 756 torch-free, where the
    #  replay measures nothing either way (765.3 against 766.1), profiles/r05e_*, r05pq_*.  The captures happen in the warm-up steps when --warmup >= 3.)
    inv_prec = {v: k for k, v in lib.PRECISIONS.items()}
    det_prec_name = inv_prec[h.cfg.det_precision - 1] if h.cfg.det_precision else a.precision
    weights.load_into(h, [hs, ys])
    log(f"rank {rank}: handle ready (batch {B}, key-points {a.precision}, detector {det_prec_name})")
    gather_used = a.gather
    if multi and a.gather == "rccl":
        try:
            shard.init_rccl(h, rank, world)
        except Exception as e:                        # labelled, never silent: reported in the JSON line
            print(f"[bench] rank {rank}: library RCCL bootstrap failed ({e}); using torch.distributed all_gather", file=sys.stderr)
            gather_used = "dist"
        flag = torch.tensor([1 if gather_used == "dist" else 0], device=tdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            gather_used = "dist"

    # this rank's contiguous chunk of the (weak-scaled) clip: K*B frames per rank (distinct content per rank), in pageable host memory
    n_local = K * B
    base = synth.clip(seed=rank, n=min(a.distinct, n_local), h=a.height, w=a.width)
    clip = np.ascontiguousarray(np.concatenate([base] * (-(-n_local // len(base))))[:n_local])
    frames = clip[:B]
    out = np.zeros(n_local, lib.RESULT_DTYPE)
    gathered = np.zeros(n_local * world, lib.RESULT_DTYPE) if multi else None
    if gathered is not None:
        gathered.view(np.uint8)[::4096] = 0          # touch the pages before the timed region

    def sync():                                        # every library call returns with its records on the host: nothing is in flight
        if dist is not None:
            dist.barrier()

    # W untimed warm-up steps.  A handle that replays its step as a hipGraph inside calls of >= 3 steps (use_graph = 2: the multi-rank path) captures its graphs in the first
    # such call (~80 ms per pipeline slot): if the W the driver asked for would leave that to the timed call, the untimed call is widened to three steps (reported as
    # config.graph_priming_steps; `warmup` in the line stays what was asked for).
    warm_steps = min(W, K)
    priming = 3 if (h.cfg.use_graph == 2 and K >= 3 and warm_steps < 3) else 0
    if max(warm_steps, priming) > 0:
        h.process(clip[:max(warm_steps, priming) * B], out[:max(warm_steps, priming) * B])
    sync()
    t0 = time.perf_counter()
    h.process(clip, out)                              # K steps of B frames, from pageable host memory (eagle_process_frames)
    t_proc = time.perf_counter()
    if multi:
        allrec = shard.gather_records(out, n_local * world, rank, world, handle=h, transport=gather_used, out=gathered, force=True)
    else:
        allrec = out
    t_gath = time.perf_counter()
    sync()
    dt = time.perf_counter() - t0
    parts = {"process_ms": round((t_proc - t0) * 1e3, 2), "gather_ms": round((t_gath - t_proc) * 1e3, 2), "barrier_ms": round((t0 + dt - t_gath) * 1e3, 2)}
    if multi:
        log(f"rank {rank}: timed region parts {parts}")
    if multi:
        tt = torch.tensor([dt], device=tdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    total_frames = n_local * world
    assert len(allrec) == total_frames
    if multi:
        lo = rank * n_local
        assert allrec[lo:lo + n_local].tobytes() == out.tobytes(), "the gathered records of this rank differ from its own"
    tm = h.timings()
    saturation = {"sat_events": int(tm.sat_events), "sat_frames": int(tm.sat_frames),
                  "note": "f32s activation stores clipped at +-4094 during the timed call (EagleTimings; a non-zero count makes the call return EAGLE_E_RANGE)"}
    log(f"timed region {dt:.3f} s -> {total_frames / dt:.1f} frames/s (host frames)")

    extras = not multi and not a.no_extras
    d_clip = h.upload(clip)                           # the clip resident in HBM: resident rate, profiling, family runs
    resident = hostsrc = None
    if extras:
        o2 = np.zeros(n_local, lib.RESULT_DTYPE)
        h.process_device(d_clip, min(2 * B, n_local), o2[:min(2 * B, n_local)])
        t1 = time.perf_counter()
        h.process_device(d_clip, n_local, o2)
        r_res = n_local / (time.perf_counter() - t1)
        assert o2.tobytes() == out.tobytes(), "resident-input records differ from host-input records"
        resident = {"value": round(r_res, 2), "unit": "frames/s", "value_over_resident": round(total_frames / dt / r_res, 4),
                    "note": "eagle_process_device_frames: the clip already in HBM when the timed region starts; records identical to the host-fed run's"}
        hp = h.host_frames(n_local)
        hp[:] = clip
        h.process(hp[:2 * B])
        t1 = time.perf_counter()
        h.process(hp)
        r_pin = n_local / (time.perf_counter() - t1)
        h.host_free(hp)
        hostsrc = {"pageable": round(total_frames / dt, 2), "pinned": round(r_pin, 2), "unit": "frames/s",
                   "note": "eagle_process_frames: pageable = value (frames copied into the library's pinned ring by "
                           f"{os.environ.get('EAGLE_COPY_THREADS', '8')} worker threads, then DMA on a copy stream under the networks of the previous batch); "
                           "pinned = frames in eagle_host_alloc memory, DMA in place"}
        log(f"resident {r_res:.1f} frames/s, pinned source {r_pin:.1f} frames/s")

    latency = None
    if extras and a.latency_calls > 0:
        # small-batch behaviour (north_star's API is Processor.process(frame); the reference's loop hands over one frame per iteration, cm.py:277):
        # per-call wall time of eagle_process_frames at B = 1, 2, 4, 8 on the default handle (small-batch mode = the library's default for batch <= EAGLE_SMALL_BATCH:
        # hipGraph replay + HRNet's branches on their own streams), and the same handle with both switched off
        fr = clip[:max(16, min(len(base), 64))]
        latency = {"unit": "ms per eagle_process_frames call of B frames from pageable host memory (records back on the host)",
                   "rows": latency_table(lib, weights, hs, ys, a, dev_index, fr, a.latency_calls,
                                         modes=[("default (small-batch mode: hipGraph + branch streams)", {}, {}),
                                                ("plain launches, one stream per network", {}, {"multi_stream": 0, "use_graph": 0})])}

    cadence = None
    if a.cadence > 0:
        from eagle_amd import clip as clipmod
        kint, hint = max(1, int(a.cadence / 3)), max(1, int(a.cadence / 1))
        st = {}
        clipmod.run_clip(h, d_clip, min(n_local, 2 * kint + 1), kint, hint, False, st)          # warm-up
        t1 = time.perf_counter()
        clipmod.run_clip(h, d_clip, n_local, kint, hint, False, st)
        dtc = time.perf_counter() - t1
        if multi:                                        # configs[4] shape: one clip per rank (the cadence is sequential within a clip), slowest rank counts
            tc = torch.tensor([dtc], device=tdev, dtype=torch.float64)
            dist.all_reduce(tc, op=dist.ReduceOp.MAX)
            dtc = float(tc.item())
        cadence = {"value": round(n_local * world / dtc, 2), "unit": "frames/s", "fps": a.cadence, "keypoint_interval": kint, "homography_interval": hint,
                   "hrnet_frames": len(st["detected_frames"]), "frames": n_local * world, "clips": world,
                   "note": "stateful reference cadence (cm.py:205-206): detector on every frame, HRNet on hrnet_frames of each clip's frames, LK flow + loop body per frame; one clip per GPU"}
        log(f"reference cadence @{a.cadence} fps: {cadence['value']} frames/s ({cadence['hrnet_frames']} HRNet frames)")

    cmc = None
    if extras:                                           # tracker side stage: boxmot's default camera-motion estimator over the whole clip (K17)
        h.clip_open(d_clip, min(n_local, 4)); h.clip_motion_ecc(0, min(n_local, 4)); h.clip_close()      # warm-up
        t1 = time.perf_counter()
        h.clip_open(d_clip, n_local)
        wm, okm = h.clip_motion_ecc(0, n_local, return_ok=True)
        dte = time.perf_counter() - t1
        t1 = time.perf_counter()
        h.clip_motion_ecc(0, n_local)                    # the 0.15-scale images exist now: the alignment launch + read-back alone
        dtk = time.perf_counter() - t1
        h.clip_close()
        cmc = {"value": round(n_local / dte, 1), "unit": "frames/s", "alignment_only": round(n_local / dtk, 1), "failed_alignments": int((okm == 0).sum()),
               "note": "eagle_clip_motion_ecc on the resident clip: gray pyramids + 0.15-scale images + one ECC workgroup per frame pair (<= 100 iterations inside the launch); "
                       "runs once per clip when track ids with camera-motion compensation are asked for, not part of value"}
        log(f"camera motion (ECC): {cmc['value']} frames/s incl. gray images, {cmc['alignment_only']} alignment only")

    # dominant kernel = the implicit-GEMM convolution family: per-launch HIP events on the launch stream
    def profile(hh, dptr, nb, prof_steps=2):
        hh.set_profiling(1)
        ms = flop = 0.0
        nc = 0
        o = np.zeros(nb, lib.RESULT_DTYPE)
        for _ in range(prof_steps):
            hh.process_device(dptr, nb, o)
            t = hh.timings()
            ms += t.conv_ms; flop += t.conv_flop; nc += t.n_conv_launches
        kt = hh.kernel_times()
        hh.set_profiling(0)
        return ms, flop, nc, kt, prof_steps
    _, _, _, ktimes, prof_steps = profile(h, d_clip, B)
    # The handle may run its two networks in two families (default: detector in exact fp32).  `roofline` is the KEY-POINT network's family (98.6 % of
    # the FLOP); the detector's launches (labels end in " d") are reported separately against their own family's peak (`detector_convs`).
    conv_ms = conv_flop = 0.0; n_conv = 0; conv_bytes = 0.0
    det_ms = det_flop = 0.0; n_det_conv = 0
    hbm_rows, conv_rows = [], []
    for name, ms, launches, nbytes, flop in ktimes:
        if (name.startswith("conv ") or name.startswith("bneck ")) and ms > 0:      # ("bneck": a whole Bottleneck of HRNet's layer 1 as one launch, csrc/bneck.hip)
            is_det = name.endswith(" d") and det_prec_name != a.precision
            peak = PEAK[det_prec_name if is_det else a.precision]
            us = ms * 1e3 / launches
            if is_det:
                det_ms += ms; det_flop += flop; n_det_conv += launches
            else:
                conv_ms += ms; conv_flop += flop; n_conv += launches; conv_bytes += nbytes
            conv_rows.append({"layer": name[5:] if name.startswith("conv ") else name, "family": det_prec_name if is_det else a.precision, "launches_per_step": launches // prof_steps, "avg_us": round(us, 2),
                              "ms_per_step": round(ms / prof_steps, 3),
                              "TFLOPs": round(flop / (ms * 1e-3) / 1e12, 1), "frac_mfma": round(flop / (ms * 1e-3) / 1e12 / peak, 4),
                              "GBps_algorithmic": round(nbytes / (ms * 1e-3) / 1e9, 1), "frac_hbm_6p3TBps": round(nbytes / (ms * 1e-3) / 1e9 / 6300.0, 4)})
            continue
        if nbytes > 0 and ms > 0:
            gbps = nbytes / (ms * 1e-3) / 1e9
            hbm_rows.append({"kernel": name, "launches_per_step": launches // prof_steps, "bytes_algorithmic_per_step": round(nbytes / prof_steps),
                             "avg_us": round(ms * 1e3 / launches, 2), "GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / 8000.0, 4)})
    achieved = conv_flop / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    dom = max((r for r in conv_rows if r["family"] == a.precision), key=lambda r: r["ms_per_step"], default=None)

    def family_run(prec, nframes, Bf, d_frames, hw=(a.height, a.width), det=a.detector, imgsz=a.imgsz, sd=None, want_records=0, det_prec=None):
        """A second handle of another precision family on frames that are already resident: frames/s (K steps of Bf in one call), the
        convolution family's HIP-event roofline, optionally the records of the first `want_records` frames.  det_prec None: detector in `prec`."""
        hf = lib.Handle(device=dev_index, frame_h=hw[0], frame_w=hw[1], det_variant=det, det_imgsz=imgsz, batch=Bf, precision=lib.PRECISIONS[prec],
                        det_precision=lib.DET_PREC_MIXED if det_prec == "mixed" else lib.PRECISIONS[det_prec or prec] + 1)
        weights.load_into(hf, sd or [hs, ys])
        nf = max(Bf, nframes // Bf * Bf)
        of = np.zeros(nf, lib.RESULT_DTYPE)
        hf.process_device(d_frames, min(2 * Bf, nf), of[:min(2 * Bf, nf)])
        t1 = time.perf_counter()
        hf.process_device(d_frames, nf, of)
        dtf = time.perf_counter() - t1
        mixed = det_prec is not None and det_prec != prec
        r = {"dtype": prec if not mixed else f"{prec} key-points + {'f32s trunk / f32 last C2f per level + Detect' if det_prec == 'mixed' else det_prec} detector", "value": round(nf / dtf, 2), "unit": "frames/s", "frames": nf, "frames_per_step": Bf}
        if not mixed:
            fms, fflop, fnc, _, fsteps = profile(hf, d_frames, Bf, 1)
            ach = fflop / (fms * 1e-3) / 1e12 if fms > 0 else 0.0
            r["roofline"] = {"bound": "mfma", "kernel": CONV_KERNEL[prec], "achieved": round(ach, 2), "peak": round(PEAK[prec], 1), "unit": "TFLOP/s",
                             "frac": round(ach / PEAK[prec], 4), "conv_ms_per_step": round(fms / fsteps, 3), "flop_per_frame": fflop / (fsteps * Bf)}
        hf.close()
        return r, of[:want_records].copy()

    def int_field_diffs(rec, ref):
        """Integer-field differences of two record arrays of the same frames (per-field counts over all frames)."""
        c = dict(frames=len(ref), hm_idx=0, n_kp=0, kp_pixels=0, n_det=0, det_cls=0, det_int_box=0, det_pitch_int=0, det_id=0, H_valid=0, dets_compared=0, det_unmatched=0)
        for g, o in zip(rec, ref):
            c["hm_idx"] += int((g["hm_idx"] != o["hm_idx"]).sum())
            c["H_valid"] += int(g["H_valid"] != o["H_valid"])
            if g["n_kp"] != o["n_kp"]:
                c["n_kp"] += 1
            else:
                k = int(o["n_kp"])
                c["kp_pixels"] += int(((g["kp"]["x"][:k] != o["kp"]["x"][:k]) | (g["kp"]["y"][:k] != o["kp"]["y"][:k]) | (g["kp"]["label"][:k] != o["kp"]["label"][:k])).sum())
            if g["n_det"] != o["n_det"]:
                c["n_det"] += 1
                continue
            k = int(o["n_det"])
            c["dets_compared"] += k
            c["det_cls"] += int((g["det"]["cls"][:k] != o["det"]["cls"][:k]).sum())
            c["det_id"] += int((g["det"]["id"][:k] != o["det"]["id"][:k]).sum())
            c["det_int_box"] += int(np.any([g["det"][f][:k] != o["det"][f][:k] for f in ("bx1", "by1", "bx2", "by2")], axis=0).sum())
            c["det_pitch_int"] += int(((g["det"]["pitch_x"][:k] != o["det"]["pitch_x"][:k]) | (g["det"]["pitch_y"][:k] != o["det"]["pitch_y"][:k])).sum())
            # order-independent view: detections of the reference whose (class, integer box) does not occur anywhere in this frame on the other side —
            # what is left of det_int_box / det_cls once near-tie confidences that swap two rows of the NMS order are taken out
            have = {(int(d["cls"]), int(d["bx1"]), int(d["by1"]), int(d["bx2"]), int(d["by2"])) for d in g["det"][:k]}
            c["det_unmatched"] += sum((int(d["cls"]), int(d["bx1"]), int(d["by1"]), int(d["bx2"]), int(d["by2"])) not in have for d in o["det"][:k])
        return c

    exact = fast = parity = cfg3 = splitdet = None
    nd = len(base)
    head_name = a.precision if det_prec_name == a.precision else f"{a.precision}+{det_prec_name}_detector"
    if extras and a.exact_frames > 0 and (a.precision, det_prec_name) != ("f32", "f32"):
        # the exact family (fp32 tensors, v_mfma_f32_16x16x4_f32 fmaf chains): the kernels whose records equal the oracle's bit for bit
        exact, rec_exact = family_run("f32", a.exact_frames, B, d_clip, want_records=nd)
        exact["note"] = "same path and clip with EAGLE_PREC_F32: records bit-identical to the CPU oracle (tests/test_gpu_pipeline.py::test_f32_path_identical_to_oracle)"
        log(f"exact family (fp32): {exact['value']} frames/s, conv {exact['roofline']['achieved']:.1f} TFLOP/s")
        parity = {"reference": "records of the exact (fp32, oracle-identical) family on the same frames", "frames": nd,
                  "default": head_name, head_name: int_field_diffs(out[:nd], rec_exact)}
    if extras and a.precision == "f32s" and det_prec_name != "f32s":
        splitdet, rec_sd = family_run("f32s", n_local, B, d_clip, want_records=nd)
        splitdet["note"] = "both networks in the split family (round 3's headline configuration): the detector's near-tie confidences may swap neighbouring ids (parity_counters.f32s)"
        log(f"split family incl. detector: {splitdet['value']} frames/s")
        if parity is not None:
            parity["f32s"] = int_field_diffs(rec_sd, rec_exact)
    if extras and a.fast_frames > 0 and a.precision != "f16":
        fast, rec_fast = family_run("f16", a.fast_frames, B, d_clip, want_records=nd)
        fast["note"] = "fp16 tensors, one fp16 MFMA per product: integer outputs are NOT guaranteed equal to the fp32 path's (see parity_counters); reported for reference, never as value"
        log(f"fast family (fp16): {fast['value']} frames/s, conv {fast['roofline']['achieved']:.1f} TFLOP/s")
        if parity is not None:
            parity["f16"] = int_field_diffs(rec_fast, rec_exact)
    if parity is not None:
        log(f"parity counters vs the exact family: {json.dumps(parity)}")
    if extras and a.cfg3_frames > 0 and (a.height, a.width, a.detector) == (720, 1280, "n"):
        # BASELINE.json configs[2]: 1920x1080 frames, the large detector at imgsz 960 (544.3 GFLOP per frame)
        B3 = 25
        n3 = max(B3, a.cfg3_frames // B3 * B3)
        yl = weights.make_yolo_state_dict("l", 0)
        base3 = synth.clip(seed=0, n=40, h=1080, w=1920)      # 40 distinct frames (10 until round 5): the mixed detector's id counters below compare all of them
        clip3 = np.concatenate([base3] * (-(-n3 // len(base3))))[:n3]
        d3 = h.upload(clip3)
        del clip3
        cfg3 = {"workload": f"{n3}-frame 1920x1080 synthetic clip, yolov8l@960 + HRNet-W48 keypoints + RANSAC homography (resident input)", "frames_per_step": B3}
        runs = [("f32s", None), ("f16", None)] if a.precision == "f32s" else [(a.precision, None)]
        if a.precision == "f32s" and det_prec_name == "f32":
            runs.insert(0, ("f32s", "f32"))              # the default configuration: 210 of the 544 GFLOP per frame in the exact family
        if a.precision == "f32s" and det_prec_name == "f32":
            runs.insert(1, ("f32s", "mixed"))            # VERDICT r5 task 7: split trunk, exact last C2f per level + Detect — measured against the id contract below
        rec3 = {}
        for pr, dp in runs:
            r3, rec3[dp or pr] = family_run(pr, n3, B3, d3, hw=(1080, 1920), det="l", imgsz=960, sd=[hs, yl], det_prec=dp, want_records=len(base3))
            cfg3["default" if dp == "f32" else "mixed_detector" if dp == "mixed" else pr] = r3
            log(f"cfg3 {r3['dtype']}: {r3['value']} frames/s")
        if "f32" in rec3:
            # integer-field differences against the default configuration (exact detector; the key-point network is the same split family in all three, so every count is the detector's)
            cfg3["parity_counters_vs_default"] = {k: int_field_diffs(rec3[k], rec3["f32"]) for k in ("mixed", "f32s") if k in rec3}
            cfg3["parity_counters_vs_default"]["frames"] = len(base3)
            cfg3["mixed_detector"]["kept"] = False
            cfg3["mixed_detector"]["note"] = ("EAGLE_DET_PREC_MIXED is selectable and NOT the default: VERDICT r5 task 7 keeps it only if every counter is 0 "
                                              "(see parity_counters_vs_default.mixed)")
            log(f"cfg3 id counters vs the exact detector: {json.dumps(cfg3['parity_counters_vs_default'])}")
        h.free(d3)

    realistic = None
    if extras and a.realistic_frames > 0 and a.precision == "f32s" and (a.height, a.width, a.detector) == (720, 1280, "n"):
        # A REALISTIC workload beside the stress one ("report separately"): `value` runs seeded random weights — noise-like heat-maps, so
        # RANSAC runs all 2000 iterations and fails on almost every frame, and ~300 detections per frame, the NMS worst case.  Here: the matched-filter key-point head of
        # tests/golden/make_peaked_head.py (geometrically consistent key-points: H is solvable, RANSAC stops early, DLT-on-inliers + LM + projection + boundaries run) and
        # the class biases of test_f32s_ids_identical_with_a_sparse_detector (a few dozen detections per frame).  Same clip, same handle configuration, same call.
        g = np.load(os.path.join(ROOT, "tests", "golden", "peaked_head.npz"))
        hs2 = dict(hs); hs2["unnormalized_model.1.weight"] = g["weight"]; hs2["unnormalized_model.1.bias"] = g["bias"]
        ys2 = dict(ys)
        for lvl in range(3):
            ys2[f"model.22.cv3.{lvl}.2.bias"] = (ys[f"model.22.cv3.{lvl}.2.bias"] - np.float32(1.25)).astype(np.float32)
        hr_ = lib.Handle(device=dev_index, frame_h=a.height, frame_w=a.width, det_variant=a.detector, det_imgsz=a.imgsz, batch=B, precision=lib.PRECISIONS[a.precision], **det_kw)
        weights.load_into(hr_, [hs2, ys2])
        nr = max(B, min(a.realistic_frames, n_local) // B * B)
        outr = np.zeros(nr, lib.RESULT_DTYPE)
        hr_.process(clip[:min(2 * B, nr)], outr[:min(2 * B, nr)])
        t1 = time.perf_counter()
        hr_.process(clip[:nr], outr)
        dtr = time.perf_counter() - t1
        _, _, _, ktr, psr = profile(hr_, d_clip, B)
        kus = {nm: round(ms * 1e3 / max(ln, 1), 1) for nm, ms, ln, _, _ in ktr if nm in ("nms", "post (geometry)", "yolo_decode")}
        _, _, _, kts, pss = profile(h, d_clip, B)
        kus_stress = {nm: round(ms * 1e3 / max(ln, 1), 1) for nm, ms, ln, _, _ in kts if nm in ("nms", "post (geometry)", "yolo_decode")}
        hr_.close()
        nd_ = min(len(base), nr)
        realistic = {"value": round(nr / dtr, 2), "unit": "frames/s", "frames": nr, "frames_per_step": B,
                     "H_valid_fraction": round(float(outr["H_valid"][:nd_].mean()), 4), "bounds_valid_fraction": round(float(outr["bounds_valid"][:nd_].mean()), 4),
                     "detections_per_frame": round(float(outr["n_det"][:nd_].mean()), 1), "candidates_per_frame": round(float(outr["n_candidates"][:nd_].mean()), 1),
                     "keypoints_per_frame": round(float(outr["n_kp"][:nd_].mean()), 1),
                     "kernel_us_per_step": kus, "kernel_us_per_step_stress_weights": kus_stress,
                     "stress_for_comparison": {"H_valid_fraction": round(float(out["H_valid"][:nd_].mean()), 4), "detections_per_frame": round(float(out["n_det"][:nd_].mean()), 1)},
                     "note": "same clip and handle configuration as `value`, but matched-filter key-point head (tests/golden/peaked_head.npz: H solvable, RANSAC stops early, LM + "
                             "projection + boundaries execute) and a sparse detector (class biases - 1.25); `value` stays the stress configuration (random heads: 2000 RANSAC "
                             "iterations without consensus per frame, ~300 boxes per frame through NMS)"}
        log(f"realistic workload: {realistic['value']} frames/s, H_valid {realistic['H_valid_fraction']}, {realistic['detections_per_frame']} detections / frame, kernels {kus} (stress: {kus_stress})")

    traffic = traffic_src = None
    tf = os.path.join(ROOT, "profiles", "conv_hbm_traffic.json" if a.precision == "f16" else f"conv_hbm_traffic_{a.precision}.json")     # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.py)
    if os.path.exists(tf):
        tj = json.load(open(tf))
        if tj.get("batch") == B and tj.get("detector") == a.detector and tj.get("precision") == a.precision:
            traffic = tj["conv_family"]["hbm_bytes_per_launch"]
            traffic_src = f"NOT measured in this run: read from profiles/{os.path.basename(tf)} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of build {tj.get('build', '?')}; the library's streams run concurrently in those passes, so kernels of other streams that overlap a convolution are counted into it: an upper bound)"
    res = None
    if rank == 0:
        dt_names = {"f16": "f16", "f32": "f32", "f32s": "f32 (split: hi/lo binary16 pairs, 3 x fp16 MFMA per product, fp32 accumulate)"}
        res = {
            "metric": f"frames/sec end-to-end (detect+keypoint+homography) @{a.width}x{a.height}",
            "value": round(total_frames / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dt_names[a.precision] + ("" if det_prec_name == a.precision else f"; detector (1.4 % of the FLOP): {dt_names[det_prec_name]}, exact v_mfma_f32_16x16x4_f32 chains"),
            "data": f"synthetic ({len(base)} distinct generated frames per rank tiled to {n_local}; seeded synthetic weights)",
            "config": {"workload": f"{n_local}-frame {a.width}x{a.height} synthetic clip per GPU, yolov8{a.detector}@{a.imgsz} + HRNet-W48 keypoints + RANSAC homography",
                       "frames_per_step": B, "frames_total": total_frames, "parallelism": f"frame-shard x{world}", "input": "pageable host memory (eagle_process_frames)",
                       "keypoint_precision": a.precision, "detector_precision": det_prec_name,
                       "gather": "none" if not multi else gather_used, "hip_graph": bool(h.cfg.use_graph == 1 or (h.cfg.use_graph == 2 and K >= 3)), "graph_priming_steps": priming, "cpu_binding_rank0": cpu_binding, "timed_region_parts_rank0": parts,
                       "runtime": "torch-bundled ROCm (torch imported before libeagle_hip.so)" if multi else "/opt/rocm (torch-free process)"},
            "roofline": {"bound": "mfma", "kernel": f"{CONV_KERNEL[a.precision]} (the {n_conv // prof_steps} convolution launches per step of the key-point network's family)",
                         "achieved": round(achieved, 2), "peak": round(PEAK[a.precision], 1), "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK[a.precision], 4),
                         "flop_per_frame": conv_flop / (prof_steps * B), "avg_launch_us": round(conv_ms * 1e3 / max(n_conv, 1), 2),
                         "conv_ms_per_step": round(conv_ms / prof_steps, 3),
                         "algorithmic_bytes_per_launch": round(conv_bytes / max(n_conv, 1)),
                         "dominant_kernel": None if dom is None else {"name": f"conv_f16_kernel / conv_split_ad_kernel instance of layer '{dom['layer']}'" if a.precision == "f32s" else dom["layer"],
                                                                      "layer": dom["layer"], "launches_per_step": dom["launches_per_step"], "avg_us": dom["avg_us"],
                                                                      "frac": dom["frac_mfma"], "ms_per_step": dom["ms_per_step"]},
                         "traffic": traffic, "traffic_source": traffic_src},
            "value_is": "frames/s with the clip in PAGEABLE HOST memory when the timed region starts (SURVEY §8d: H2D -> networks -> geometry -> records on the host); "
                        "resident.value is the same path with the clip already in HBM",
            "saturation": saturation,
        }
        if det_prec_name != a.precision and det_ms > 0:
            res["detector_convs"] = {"family": det_prec_name, "launches_per_step": n_det_conv // prof_steps, "ms_per_step_serialised": round(det_ms / prof_steps, 3),
                                     "achieved": round(det_flop / (det_ms * 1e-3) / 1e12, 2), "peak": PEAK[det_prec_name], "unit": "TFLOP/s",
                                     "frac": round(det_flop / (det_ms * 1e-3) / 1e12 / PEAK[det_prec_name], 4), "flop_per_frame": det_flop / (prof_steps * B),
                                     "note": "runs on its own stream beside the key-point network in the timed region; serialised here by the profiling mode"}
        if resident is not None:
            res["resident"] = resident
        if hostsrc is not None:
            res["host_sources"] = hostsrc
        if hbm_rows:
            res["roofline_hbm"] = hbm_rows
        if conv_rows:      # the ten convolution layer shapes that take the most time, each against BOTH roofs (MFMA peak of its family; 6.3 TB/s achievable HBM)
            res["roofline_conv_layers"] = sorted(conv_rows, key=lambda r: -r["ms_per_step"])[:(len(conv_rows) if a.all_layers else 10)]
        if splitdet is not None:
            res["split_detector"] = splitdet
        if fast is not None:
            res["fast_family"] = fast
        if exact is not None:
            res["exact_family"] = exact
        if parity is not None:
            res["parity_counters"] = parity
        if cfg3 is not None:
            res["cfg3"] = cfg3
        if realistic is not None:
            res["realistic"] = realistic
        if latency is not None:
            res["latency"] = latency
        if cadence is not None:
            res["reference_cadence"] = cadence
        if cmc is not None:
            res["camera_motion_ecc"] = cmc
        if cpu_base is not None:
            res["cpu_baseline"] = cpu_base
    h.free(d_clip)
    h.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        os.write(result_fd, (json.dumps(res) + "\n").encode())


if __name__ == "__main__":
    main()

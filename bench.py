#!/usr/bin/env python3
"""bench.py — frames/sec of the per-frame path (detect + keypoints + homography) at 1280x720.

One "step" = one pass of the hot path over one device batch of ``--batch`` synthetic frames that are already
resident in HBM (C ABI: eagle_process_device_frames).  Default run = BASELINE.json configs[1]: a 1000-frame
1280x720 synthetic clip, YOLOv8-n detector + HRNet-W48 keypoint model, one MI355X (20 steps x 50 frames).
The K timed steps are issued as ONE library call over the K*batch-frame clip so that the library's two-deep pipeline
(geometry + record copy of step i under the networks of step i+1) is part of what is measured.
Multi-GPU (driver launches one rank per GPU through torch.distributed.run): frames shard by contiguous chunk,
weights replicated, no data-path collective; ONE all-gather of the fixed-size records at the end (inside the timed
region); value = total frames of all ranks / max-over-ranks time  ("scaling": "weak").

Default family = EAGLE_PREC_F32S ("f32s": fp32-grade results — records equal the fp32 oracle's, tests/test_gpu_pipeline.py::test_f32s_* —
computed as three fp16 MFMAs per product over (hi, lo) binary16 tensors).  ``value`` = frames/s with the clip already resident in HBM (the
measurement contract's definition).  The same line also carries ``pcie_inclusive`` (SURVEY §8d's definition: the frames start in HOST
memory; pageable memory through the library's pinned ring, and pinned memory), ``fast_family`` (the fp16 family on the same clip),
``exact_family`` (the bit-exact fp32 family on the same clip), ``parity_counters`` (integer-field differences of both faster families against
the exact family's records on the distinct frames of the clip — GPU against GPU, the exact family being the one that equals the oracle bit
for bit), ``cfg3`` (BASELINE configs[2]: 1920x1080, yolov8l@960, a 200-frame run of both families), ``roofline`` (MFMA, the convolution
family) and ``roofline_hbm`` (the bandwidth-bound kernels against 8 TB/s, HIP events on their launch streams / algorithmic bytes).

Prints ONE JSON line on rank 0.  The CPU oracle appears here only as the timed ``cpu_baseline`` leg.  At N = 1 torch is not imported
before the GPU work (the library has its own streams and synchronises its calls itself)."""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MFMA_PEAK_TFLOPS = 2500.0         # dense fp16/bf16 MFMA peak, MI355X_MICROARCH.md
# peak of the convolution family per precision, in ALGORITHMIC TFLOP/s (2 x MAC of the convolution): the split family spends three fp16 MFMA
# products per algorithmic product (hi*hi + hi*lo + lo*hi), so its roof is a third of the dense fp16 MFMA peak
PEAK = {"f16": MFMA_PEAK_TFLOPS, "f32": 157.3, "f32s": MFMA_PEAK_TFLOPS / 3.0}
CONV_KERNEL = {"f16": "conv_f16_ad_kernel / conv_f16_ws_kernel / conv_f16_kernel", "f32": "conv_f32_kernel",
               "f32s": "conv_f16_kernel<..., SPLIT> family (3 x v_mfma_f32_16x16x32_f16 per K-step)"}


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
                    help="f32s (default): split-precision family, fp32-grade results (records equal the fp32 oracle's); f16: the fast family; f32: the bit-exact family")
    ap.add_argument("--distinct", type=int, default=20, help="distinct synthetic frames generated (tiled to the clip)")
    ap.add_argument("--cpu-frames", type=int, default=40)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the network phase as a hipGraph (no gain once a step is GPU-bound)")
    ap.add_argument("--host-frames", action="store_true", help="(kept for compatibility: the PCIe-inclusive path is always timed at N = 1)")
    ap.add_argument("--all-layers", action="store_true", help="roofline_conv_layers lists every convolution layer shape instead of the ten heaviest")
    ap.add_argument("--no-extras", action="store_true", help="skip pcie_inclusive / exact_family / roofline_hbm (profiling runs)")
    ap.add_argument("--exact-frames", type=int, default=1000, help="frames of the fp32 exact-family run (0: skip)")
    ap.add_argument("--fast-frames", type=int, default=1000, help="frames of the fp16 fast-family run (0: skip)")
    ap.add_argument("--cfg3-frames", type=int, default=200, help="frames of the configs[2] run (1920x1080, yolov8l@960; 0: skip)")
    ap.add_argument("--cadence", type=int, default=0, metavar="FPS", help="also time the reference's default cadence on the same clip: get_coordinates(frames, FPS, num_homography=1, "
                    "num_keypoint_detection=3) = HRNet every int(FPS/3)-th frame, optical-flow propagation in between (stateful; reported as reference_cadence, never as value)")
    ap.add_argument("--gather", default="rccl", choices=["rccl", "dist"])
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"], help="process-group backend (gloo + --shared-gpu: dev test of the multi-rank path on one GPU)")
    ap.add_argument("--shared-gpu", action="store_true", help="every rank uses HIP device 0 (developer test only)")
    a = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")

    dist = torch = None
    dev_index = 0 if a.shared_gpu else local_rank
    tdev = "cuda" if a.backend == "nccl" else "cpu"
    if world > 1:
        import torch
        import torch.distributed as dist
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
    h = lib.Handle(device=dev_index, frame_h=a.height, frame_w=a.width, det_variant=a.detector, det_imgsz=a.imgsz,
                   batch=B, precision=lib.PRECISIONS[a.precision],
                   use_graph=1 if a.graph else 0)
    weights.load_into(h, [hs, ys])
    log(f"rank {rank}: handle ready (batch {B})")
    gather_used = a.gather
    if world > 1 and a.gather == "rccl":
        try:
            shard.init_rccl(h, rank, world)
        except Exception as e:                        # labelled, never silent: reported in the JSON line
            print(f"[bench] rank {rank}: library RCCL bootstrap failed ({e}); using torch.distributed all_gather", file=sys.stderr)
            gather_used = "dist"
        flag = torch.tensor([1 if gather_used == "dist" else 0], device=tdev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()):
            gather_used = "dist"

    # this rank's contiguous chunk of the (weak-scaled) clip: K*B frames per rank (distinct content per rank), resident in HBM
    n_local = K * B
    base = synth.clip(seed=rank, n=min(a.distinct, n_local), h=a.height, w=a.width)
    clip = np.concatenate([base] * (-(-n_local // len(base))))[:n_local]
    frames = clip[:B]
    d_clip = h.upload(clip)                           # inputs resident in HBM before the timed region
    log(f"rank {rank}: {n_local} frames resident in HBM")
    out = np.zeros(n_local, lib.RESULT_DTYPE)
    gathered = np.zeros(n_local * world, lib.RESULT_DTYPE) if world > 1 else None
    if gathered is not None:
        gathered.view(np.uint8)[::4096] = 0          # touch the pages before the timed region

    def sync():                                        # every library call returns with its records on the host: nothing is in flight
        if dist is not None:
            dist.barrier()

    if W > 0:
        h.process_device(d_clip, min(W, K) * B, out[:min(W, K) * B])
    sync()
    t0 = time.perf_counter()
    h.process_device(d_clip, n_local, out)            # K steps of B frames
    if world > 1:
        allrec = shard.gather_records(out, n_local * world, rank, world, handle=h, transport=gather_used, out=gathered)
    else:
        allrec = out
    sync()
    dt = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([dt], device=tdev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    total_frames = n_local * world
    assert len(allrec) == total_frames
    log(f"timed region {dt:.3f} s -> {total_frames / dt:.1f} frames/s")

    extras = world == 1 and not a.no_extras
    pcie = None
    if extras:
        # SURVEY §8d's metric: the clip starts in host memory.  (a) pageable numpy memory -> worker threads -> pinned ring -> DMA;
        # (b) pinned memory (eagle_host_alloc: where a decoder would write its frames) -> DMA in place.
        h.process(clip[:2 * B])
        t1 = time.perf_counter()
        h.process(clip)
        r_page = n_local / (time.perf_counter() - t1)
        hp = h.host_frames(n_local)
        hp[:] = clip
        h.process(hp[:2 * B])
        t1 = time.perf_counter()
        h.process(hp)
        r_pin = n_local / (time.perf_counter() - t1)
        h.host_free(hp)
        pcie = {"value": round(r_page, 2), "unit": "frames/s", "frac_of_resident": round(r_page / (total_frames / dt), 4),
                "pinned_source": round(r_pin, 2), "pinned_frac_of_resident": round(r_pin / (total_frames / dt), 4),
                "note": "eagle_process_frames: value = frames in pageable host memory (copied into the library's pinned ring by "
                        f"{os.environ.get('EAGLE_COPY_THREADS', '8')} worker threads, then DMA on a copy stream under the networks of the previous batch); "
                        "pinned_source = frames in eagle_host_alloc memory, DMA in place"}
        log(f"PCIe-inclusive: {r_page:.1f} frames/s from pageable memory, {r_pin:.1f} from pinned memory")

    cadence = None
    if a.cadence > 0:
        from eagle_amd import clip as clipmod
        kint, hint = max(1, int(a.cadence / 3)), max(1, int(a.cadence / 1))
        st = {}
        clipmod.run_clip(h, d_clip, min(n_local, 2 * kint + 1), kint, hint, False, st)          # warm-up
        t1 = time.perf_counter()
        clipmod.run_clip(h, d_clip, n_local, kint, hint, False, st)
        dtc = time.perf_counter() - t1
        if world > 1:                                    # configs[4] shape: one clip per rank (the cadence is sequential within a clip), slowest rank counts
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
    conv_ms, conv_flop, n_conv, ktimes, prof_steps = profile(h, d_clip, B)
    achieved = conv_flop / (conv_ms * 1e-3) / 1e12 if conv_ms > 0 else 0.0
    # bandwidth-bound kernels: algorithmic bytes (inputs once + outputs once, counted by the library per launch) / HIP-event time
    hbm_rows, conv_rows = [], []
    conv_bytes = sum(nbytes for name, _, _, nbytes, _ in ktimes if name.startswith("conv "))
    for name, ms, launches, nbytes, flop in ktimes:
        if name.startswith("conv ") and ms > 0:
            us = ms * 1e3 / launches
            conv_rows.append({"layer": name[5:], "launches_per_step": launches // prof_steps, "avg_us": round(us, 2), "ms_per_step": round(ms / prof_steps, 3),
                              "TFLOPs": round(flop / (ms * 1e-3) / 1e12, 1), "frac_mfma": round(flop / (ms * 1e-3) / 1e12 / PEAK[a.precision], 4),
                              "GBps_algorithmic": round(nbytes / (ms * 1e-3) / 1e9, 1), "frac_hbm_6p3TBps": round(nbytes / (ms * 1e-3) / 1e9 / 6300.0, 4)})
            continue
        if nbytes > 0 and ms > 0:
            gbps = nbytes / (ms * 1e-3) / 1e9
            hbm_rows.append({"kernel": name, "launches_per_step": launches // prof_steps, "bytes_algorithmic_per_step": round(nbytes / prof_steps),
                             "avg_us": round(ms * 1e3 / launches, 2), "GBps": round(gbps, 1), "frac_of_8TBps": round(gbps / 8000.0, 4)})

    def family_run(prec, nframes, Bf, d_frames, hw=(a.height, a.width), det=a.detector, imgsz=a.imgsz, sd=None, want_records=0, det_prec=None):
        """A second handle of another precision family on frames that are already resident: frames/s (K steps of Bf in one call), the
        convolution family's HIP-event roofline, optionally the records of the first `want_records` frames."""
        hf = lib.Handle(device=dev_index, frame_h=hw[0], frame_w=hw[1], det_variant=det, det_imgsz=imgsz, batch=Bf, precision=lib.PRECISIONS[prec],
                        det_precision=0 if det_prec is None else lib.PRECISIONS[det_prec] + 1)
        weights.load_into(hf, sd or [hs, ys])
        nf = max(Bf, nframes // Bf * Bf)
        of = np.zeros(nf, lib.RESULT_DTYPE)
        hf.process_device(d_frames, min(2 * Bf, nf), of[:min(2 * Bf, nf)])
        t1 = time.perf_counter()
        hf.process_device(d_frames, nf, of)
        dtf = time.perf_counter() - t1
        fms, fflop, fnc, _, fsteps = profile(hf, d_frames, Bf, 1)
        hf.close()
        ach = fflop / (fms * 1e-3) / 1e12 if fms > 0 else 0.0
        r = {"dtype": prec, "value": round(nf / dtf, 2), "unit": "frames/s", "frames": nf, "frames_per_step": Bf,
             "roofline": {"bound": "mfma", "kernel": CONV_KERNEL[prec], "achieved": round(ach, 2), "peak": round(PEAK[prec], 1), "unit": "TFLOP/s",
                          "frac": round(ach / PEAK[prec], 4), "conv_ms_per_step": round(fms / fsteps, 3), "flop_per_frame": fflop / (fsteps * Bf)}}
        return r, of[:want_records].copy()

    def int_field_diffs(rec, ref):
        """Integer-field differences of two record arrays of the same frames (per-field counts over all frames)."""
        c = dict(frames=len(ref), hm_idx=0, n_kp=0, kp_pixels=0, n_det=0, det_cls=0, det_int_box=0, det_pitch_int=0, H_valid=0, dets_compared=0, det_unmatched=0)
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
            c["det_int_box"] += int(np.any([g["det"][f][:k] != o["det"][f][:k] for f in ("bx1", "by1", "bx2", "by2")], axis=0).sum())
            c["det_pitch_int"] += int(((g["det"]["pitch_x"][:k] != o["det"]["pitch_x"][:k]) | (g["det"]["pitch_y"][:k] != o["det"]["pitch_y"][:k])).sum())
            # order-independent view: detections of the reference whose (class, integer box) does not occur anywhere in this frame on the other side —
            # what is left of det_int_box / det_cls once near-tie confidences that swap two rows of the NMS order are taken out
            have = {(int(d["cls"]), int(d["bx1"]), int(d["by1"]), int(d["bx2"]), int(d["by2"])) for d in g["det"][:k]}
            c["det_unmatched"] += sum((int(d["cls"]), int(d["bx1"]), int(d["by1"]), int(d["bx2"]), int(d["by2"])) not in have for d in o["det"][:k])
        return c

    exact = fast = parity = cfg3 = None
    nd = len(base)
    if extras and a.exact_frames > 0 and a.precision != "f32":
        # the exact family (fp32 tensors, v_mfma_f32_16x16x4_f32 fmaf chains): the kernels whose records equal the oracle's bit for bit
        exact, rec_exact = family_run("f32", a.exact_frames, B, d_clip, want_records=nd)
        exact["note"] = "same path and clip with EAGLE_PREC_F32: records bit-identical to the CPU oracle (tests/test_gpu_pipeline.py::test_f32_path_identical_to_oracle)"
        log(f"exact family (fp32): {exact['value']} frames/s, conv {exact['roofline']['achieved']:.1f} TFLOP/s")
        parity = {"reference": "records of the exact (fp32, oracle-identical) family on the same frames", "frames": nd,
                  a.precision: int_field_diffs(out[:nd], rec_exact)}
    if extras and a.fast_frames > 0 and a.precision != "f16":
        fast, rec_fast = family_run("f16", a.fast_frames, B, d_clip, want_records=nd)
        fast["note"] = "fp16 tensors, one fp16 MFMA per product: integer outputs are NOT guaranteed equal to the fp32 path's (see parity_counters); reported for reference, never as value"
        log(f"fast family (fp16): {fast['value']} frames/s, conv {fast['roofline']['achieved']:.1f} TFLOP/s")
        # the mixed handle (EagleConfig::det_precision): key-points in fp16, the detector (1.4 % of the FLOPs) in the split family
        mixed, rec_mixed = family_run("f16", a.fast_frames, B, d_clip, want_records=nd, det_prec="f32s")
        fast["with_f32s_detector"] = {"value": mixed["value"], "unit": "frames/s", "note": "fp16 HRNet + f32s YOLOv8: boxes / confidences / NMS order / ids at fp32 grade"}
        log(f"fast family with the detector in f32s: {mixed['value']} frames/s")
        if parity is not None:
            parity["f16"] = int_field_diffs(rec_fast, rec_exact)
            parity["f16_with_f32s_detector"] = int_field_diffs(rec_mixed, rec_exact)
    if parity is not None:
        log(f"parity counters vs the exact family: {json.dumps(parity)}")
    if extras and a.cfg3_frames > 0 and (a.height, a.width, a.detector) == (720, 1280, "n"):
        # BASELINE.json configs[2]: 1920x1080 frames, the large detector at imgsz 960 (544.3 GFLOP per frame)
        B3 = 25
        n3 = max(B3, a.cfg3_frames // B3 * B3)
        yl = weights.make_yolo_state_dict("l", 0)
        base3 = synth.clip(seed=0, n=10, h=1080, w=1920)
        clip3 = np.concatenate([base3] * (-(-n3 // len(base3))))[:n3]
        d3 = h.upload(clip3)
        cfg3 = {"workload": f"{n3}-frame 1920x1080 synthetic clip, yolov8l@960 + HRNet-W48 keypoints + RANSAC homography", "frames_per_step": B3}
        for pr in (a.precision, "f16") if a.precision != "f16" else ("f16",):
            r3, _ = family_run(pr, n3, B3, d3, hw=(1080, 1920), det="l", imgsz=960, sd=[hs, yl])
            cfg3[pr] = r3
            log(f"cfg3 {pr}: {r3['value']} frames/s, conv {r3['roofline']['achieved']:.1f} TFLOP/s")
        h.free(d3)

    traffic = traffic_src = None
    tf = os.path.join(ROOT, "profiles", "conv_hbm_traffic.json" if a.precision == "f16" else f"conv_hbm_traffic_{a.precision}.json")     # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (tools/pmc_traffic.py)
    if os.path.exists(tf):
        tj = json.load(open(tf))
        if tj.get("batch") == B and tj.get("detector") == a.detector and tj.get("precision") == a.precision:
            traffic = tj["conv_family"]["hbm_bytes_per_launch"]
            traffic_src = f"NOT measured in this run: read from profiles/{os.path.basename(tf)} (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of build {tj.get('build', '?')}; the library's streams run concurrently in those passes, so kernels of other streams that overlap a convolution are counted into it: an upper bound)"
    res = None
    if rank == 0:
        res = {
            "metric": f"frames/sec end-to-end (detect+keypoint+homography) @{a.width}x{a.height}",
            "value": round(total_frames / dt, 2), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
            "ms_per_step": round(dt / K * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f16": "f16", "f32": "f32", "f32s": "f32 (split: hi/lo binary16 pairs, 3 x fp16 MFMA per product, fp32 accumulate)"}[a.precision], "data": f"synthetic ({len(base)} distinct generated frames per rank tiled to {n_local}; seeded synthetic weights)",
            "config": {"workload": f"{n_local}-frame {a.width}x{a.height} synthetic clip per GPU, yolov8{a.detector}@{a.imgsz} + HRNet-W48 keypoints + RANSAC homography",
                       "frames_per_step": B, "frames_total": total_frames, "parallelism": f"frame-shard x{world}",
                       "gather": "none" if world == 1 else gather_used, "hip_graph": bool(a.graph)},
            "roofline": {"bound": "mfma", "kernel": f"{CONV_KERNEL[a.precision]} (all {n_conv // prof_steps} convolution launches of a step)",
                         "achieved": round(achieved, 2), "peak": round(PEAK[a.precision], 1), "unit": "TFLOP/s",
                         "frac": round(achieved / PEAK[a.precision], 4),
                         "flop_per_frame": conv_flop / (prof_steps * B), "avg_launch_us": round(conv_ms * 1e3 / max(n_conv, 1), 2),
                         "conv_ms_per_step": round(conv_ms / prof_steps, 3),
                         "algorithmic_bytes_per_launch": round(conv_bytes / max(n_conv, 1)),
                         "traffic": traffic, "traffic_source": traffic_src},
            "value_is": "frames/s with the clip resident in HBM before the timed region (records land on the host inside it); pcie_inclusive.value is the same path fed from host memory",
        }
        if hbm_rows:
            res["roofline_hbm"] = hbm_rows
        if conv_rows:      # the ten convolution layer shapes that take the most time, each against BOTH roofs (MFMA peak; 6.3 TB/s achievable HBM)
            res["roofline_conv_layers"] = sorted(conv_rows, key=lambda r: -r["ms_per_step"])[:(len(conv_rows) if a.all_layers else 10)]
        if fast is not None:
            res["fast_family"] = fast
        if exact is not None:
            res["exact_family"] = exact
        if parity is not None:
            res["parity_counters"] = parity
        if cfg3 is not None:
            res["cfg3"] = cfg3
        if cadence is not None:
            res["reference_cadence"] = cadence
        if pcie is not None:
            res["pcie_inclusive"] = pcie
        if cmc is not None:
            res["camera_motion_ecc"] = cmc
        if not a.no_cpu_baseline and world == 1:       # rank 0 at N = 1 only (the measurement contract)
            res["cpu_baseline"] = cpu_baseline(hs, ys, frames, a.cpu_frames, max(1, min(16, usable_cpus())), variant=a.detector, imgsz=a.imgsz)
            # (all 256 hardware threads of the GPU box were tried once: torch-CPU convolutions collapse to 0.004 frames/s, 247 s for one
            #  frame, profiles/r02b_bench_default_1gpu.json, so the bounded sample stays at 16 threads and says so)
    h.free(d_clip)
    h.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(res), flush=True)


if __name__ == "__main__":
    main()

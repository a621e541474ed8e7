"""Minimal command line for the plumbing configuration (BASELINE.json configs[0]): run a clip through
the GPU path and write ``raw_coordinates.json`` exactly the way the reference's ``main.py:26-30`` does
(``json.dump(coordinates, f, default=float)`` of ``CoordinateModel.get_coordinates``; schema ``docs/data.md:20-41``).

    python -m eagle_amd.cli --frames 10 --fps 5 --out output/synthetic          # synthetic clip (no video decode here)
    python -m eagle_amd.cli --clip frames.npy --fps 25 --out output/myclip       # uint8 [n,h,w,3] BGR frames

Video decode/encode, the pandas post-processor and the annotated video of ``main.py:34-81`` are out of scope
(SURVEY §8f rows 3-4).  The cadence is main.py:27's by default (homography once per second, key-point model three times per
second, optical flow in between); ``--every-frame`` selects the stateless configuration (both on every frame)."""
import argparse
import json
import os
import time

import numpy as np


def load_state_dict(path):
    """A checkpoint file -> {name: float32 ndarray}.  torch is used only here, to read the file (weight loading is the one place the
    north star allows it)."""
    import torch
    try:                                    # plain state-dict files (.pth) need no unpickling of arbitrary objects
        obj = torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:                  # ultralytics .pt checkpoints pickle their model classes: full unpickling executes code from the file
        import warnings
        warnings.warn(f"{path}: not loadable with weights_only=True ({type(e).__name__}); falling back to full unpickling — only do this with "
                      "checkpoints you trust", stacklevel=2)
        obj = torch.load(path, map_location="cpu", weights_only=False)
    if isinstance(obj, dict) and "model" in obj and hasattr(obj["model"], "state_dict"):      # ultralytics checkpoint
        obj = obj["model"].float().state_dict()
    elif hasattr(obj, "state_dict"):
        obj = obj.state_dict()
    elif isinstance(obj, dict) and "state_dict" in obj:
        obj = obj["state_dict"]
    return {k: v.detach().float().cpu().numpy() for k, v in obj.items() if hasattr(v, "detach") and v.ndim >= 0 and not k.endswith("num_batches_tracked")}


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--clip", help=".npy file with uint8 [n,h,w,3] BGR frames (default: synthetic clip)")
    ap.add_argument("--frames", type=int, default=10)
    ap.add_argument("--fps", type=int, default=5)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--out", default="output/synthetic")
    ap.add_argument("--detector", default="n")
    ap.add_argument("--imgsz", type=int, default=640)
    ap.add_argument("--precision", default="f32s", choices=["f16", "f32", "f32s"],
                    help="f32s (default): fp32-grade results on fp16 MFMAs (what the reference's .float() path computes); f16: fastest; f32: bit-exact fp32 MFMA")
    ap.add_argument("--detector-precision", default=None, choices=["f16", "f32", "f32s"],
                    help="family of the detector alone (default: the library's — exact fp32 next to f32s key-points, so that boxes / confidences / ids are the fp32 arithmetic's bit for bit)")
    ap.add_argument("--allow-saturation", action="store_true",
                    help="f32s stores activations with a range of +-4094; by default a run in which one was clipped fails (EAGLE_E_RANGE). With this flag it only warns "
                         "(for a checkpoint with larger activations prefer --precision f32)")
    ap.add_argument("--batch", type=int, default=10)
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--num-homography", type=int, default=1, help="homography solves per second (main.py:27: 1)")
    ap.add_argument("--num-keypoint-detection", type=int, default=3, help="key-point model runs per second (main.py:27: 3)")
    ap.add_argument("--every-frame", action="store_true", help="key-points and homography on every frame (stateless configuration)")
    ap.add_argument("--letterbox", default="rect", choices=["rect", "square"],
                    help="detector input geometry: 'rect' = ultralytics LetterBox(auto=True), what the reference's .pt detectors run with (cm.py:56-57); 'square' = auto=False, the "
                         "static imgsz x imgsz input of its exported ONNX detector (the CPU default, cm.py:54-55)")
    ap.add_argument("--calibration", action="store_true")
    ap.add_argument("--tracker", action="store_true", help="key players by track id (BoT-SORT association) instead of the detection index")
    ap.add_argument("--reid", action="store_true", help="with --tracker: appearance matching with OSNet-x0.25 embeddings, as the reference configures BotSort (cm.py:66-72)")
    ap.add_argument("--reid-weights", help="with --reid: torchreid osnet_x0_25 state-dict (.pth; keys conv1.* ... fc.*, the 'reid.' prefix is added here)")
    ap.add_argument("--camera-motion", nargs="?", const="ecc", default=None, choices=["ecc", "sparse"],
                    help="with --tracker: compensate camera motion; 'ecc' (default when the flag is given) is boxmot's default estimator, i.e. the reference's "
                         "configuration; 'sparse' = BoT-SORT's sparse-optical-flow alternative on a fixed grid")
    ap.add_argument("--keypoint-weights", help="HRNet state-dict (.pth as the reference loads at cm.py:58-59: keys unnormalized_model.0.* / unnormalized_model.1.*)")
    ap.add_argument("--detector-weights", help="detector checkpoint: a torch state-dict (.pth) with ultralytics key names model.N.*, or an ultralytics .pt whose 'model' entry has .state_dict()")
    ap.add_argument("--synthetic-weights", action="store_true", help="run with seeded RANDOM networks (plumbing / benchmarking only: the coordinates are meaningless)")
    ap.add_argument("--native-fps", type=float, default=None, help="frame rate of --clip: sample it down to --fps the way read_video does (io.py:17-25)")
    a = ap.parse_args(argv)

    from . import synth
    from .coordinate_model import CoordinateModel
    frames = np.load(a.clip) if a.clip else synth.clip(a.seed, a.frames)
    if a.native_fps is not None:
        from . import io as eio
        frames, _ = eio.read_clip(frames, a.native_fps, a.fps)
    n, h, w, _ = frames.shape
    hs = ys = None
    if a.keypoint_weights:
        hs = load_state_dict(a.keypoint_weights)
    if a.detector_weights:
        ys = load_state_dict(a.detector_weights)
    if (hs is None or ys is None) and not a.synthetic_weights:
        raise SystemExit("no checkpoints given: pass --keypoint-weights and --detector-weights (the reference's keypoints_main.pth / detector_*.pt, cm.py:54-59), "
                         "or --synthetic-weights to run seeded random networks on purpose")
    if hs is None or ys is None:
        print("WARNING: running with seeded RANDOM network weights: the output has the reference's schema but no meaning", flush=True)
    model = CoordinateModel(frame_hw=(h, w), detector=a.detector, det_imgsz=a.imgsz, letterbox=a.letterbox, batch=min(a.batch, max(n, 1)),
                            precision=a.precision, detector_precision=a.detector_precision, allow_saturation=a.allow_saturation, device=a.device, seed=a.seed,
                            hrnet_state_dict=hs, detector_state_dict=ys, tracker=a.tracker, camera_motion=a.camera_motion or False,
                            reid=a.reid, reid_state_dict=({("reid." + k): v for k, v in load_state_dict(a.reid_weights).items()} if a.reid_weights else None))
    t0 = time.perf_counter()
    nh, nk = (a.fps, a.fps) if a.every_frame else (a.num_homography, a.num_keypoint_detection)
    from . import lib
    try:
        coordinates = model.get_coordinates(frames, a.fps, num_homography=nh, num_keypoint_detection=nk, verbose=False, calibration=a.calibration)
    except lib.EagleRangeError as e:
        raise SystemExit(f"error: {e}\n(re-run with --precision f32, or with --allow-saturation to accept clipped activations)")
    dt = time.perf_counter() - t0
    sat = model.handle.timings()
    if sat.sat_events:
        print(f"WARNING: {sat.sat_events} activation stores in {sat.sat_frames} frame(s) were clipped at +-4094 (f32s range); the affected frames are not fp32-grade", flush=True)
    os.makedirs(a.out, exist_ok=True)
    with open(os.path.join(a.out, "raw_coordinates.json"), "w") as f:
        json.dump(coordinates, f, default=float)
    with open(os.path.join(a.out, "metadata.json"), "w") as f:
        json.dump({"fps": a.fps, "frames": n, "seconds": dt, "note": "team_mapping needs the post-processor (out of scope)"}, f)
    print(f"{n} frames in {dt:.3f} s -> {os.path.join(a.out, 'raw_coordinates.json')}")
    return 0


if __name__ == "__main__":
    raise SystemExit(main())

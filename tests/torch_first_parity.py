#!/usr/bin/env python3
"""Parity ON THE RUNTIME THE N > 1 BENCH USES (VERDICT r4 task 1a).

At WORLD_SIZE > 1 bench.py imports torch BEFORE libeagle_hip.so, so the dynamic loader binds the library (and the RCCL it dlopens) to the ROCm runtime
bundled in the PyTorch wheel (torch/lib/libamdhip64.so, libhsa-runtime64.so, librccl.so: HIP 7.0.2) instead of /opt/rocm's 7.2 copies under which the
rest of the GPU suite runs.  This script is started as a FRESH process by tests/test_gpu_edges.py:
  1. import torch first, then the library; create the default handle;
  2. read /proc/self/maps: every mapped libamdhip64 / libhsa-runtime64 / librccl must be ONE file each, and torch's copy;
  3. run the default handle on the five cfg-2 frames and compare every record with the fp32 CPU oracle (OracleModel(backend="c")) through
     tests/test_gpu_pipeline.py::_default_handle_parity — the exception-free comparison of the default-handle tests;
  4. world-1 RCCL communicator + eagle_gather of those records (the library's RCCL is torch's copy here), bytes identical.
Prints TORCH_FIRST_PARITY_OK and the mapped runtime paths."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))            # (test infrastructure: lives under tests/ because it calls the oracle)


def mapped(pattern):
    out = set()
    with open("/proc/self/maps") as f:
        for ln in f:
            p = ln.split()[-1]
            if pattern in os.path.basename(p):
                out.add(os.path.realpath(p))
    return sorted(out)


def main():
    import torch                                            # FIRST: the order of bench.py's multi-rank path
    import numpy as np
    from eagle_amd import lib, synth, weights
    from eagle_amd.coordinate_model import CoordinateModel
    from oracle import pipeline
    from test_gpu_pipeline import _default_handle_parity

    hs, ys = weights.make_hrnet_state_dict(0), weights.make_yolo_state_dict("n", 0)
    frames = np.stack([synth.frame(0, 0), synth.frame(0, 37), synth.noise_frame(1), synth.frame(0, 12), synth.frame(2, 5)])
    cm = CoordinateModel(batch=2, hrnet_state_dict=hs, detector_state_dict=ys)          # all defaults: f32s key-points, exact-fp32 detector
    assert cm.handle.cfg.precision == lib.PREC_F32S and cm.handle.cfg.det_precision == lib.PREC_F32 + 1
    recs = cm.process_records(frames)

    # the communicator + gather on the same runtime (world 1: RCCL cannot place two ranks on one device)
    cm.handle.comm_init(0, 1, lib.comm_unique_id())
    g = cm.handle.gather(recs, 1)
    assert g.tobytes() == np.ascontiguousarray(recs).tobytes(), "eagle_gather changed the records"

    tlib = os.path.realpath(os.path.join(os.path.dirname(torch.__file__), "lib"))
    report = {}
    for name in ("libamdhip64", "libhsa-runtime64", "librccl"):
        paths = mapped(name)
        report[name] = paths
        assert len(paths) == 1, f"{name}: {len(paths)} copies mapped: {paths}"
        assert paths[0].startswith(tlib + os.sep), f"{name} is not torch's copy: {paths[0]} (torch/lib = {tlib})"
    print("mapped runtime:", report, "torch", torch.__version__, "hip", torch.version.hip, flush=True)

    ora = pipeline.OracleModel(hs, ys, backend="c")
    nd = []
    for i, f in enumerate(frames):
        oref, aux = ora.step(f, i)
        nd.append(_default_handle_parity(recs[i], oref, aux, f"torch-first default handle frame {i}", (720, 1280)))
    cm.handle.close()
    assert sum(nd) > 1000, nd
    print(f"TORCH_FIRST_PARITY_OK detections={nd}", flush=True)


if __name__ == "__main__":
    main()

"""Do two independent pipelines on one GPU (two handles, two host threads, half of the clip each) finish sooner than one handle over the whole clip?
If the HBM-bound layers of one batch can hide under the MFMA-bound layers of another, they do; under a pure power cap they do not.
usage: python tools/probes/two_handles_probe.py [batch]"""
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from eagle_amd import lib, synth, weights

B = int(sys.argv[1]) if len(sys.argv) > 1 else 50
hs, ys = weights.make_hrnet_state_dict(0), weights.make_yolo_state_dict("n", 0)
base = synth.clip(seed=0, n=20)
clip = np.ascontiguousarray(np.concatenate([base] * 50))        # 1000 frames


def handle(b):
    h = lib.Handle(batch=b)
    weights.load_into(h, [hs, ys])
    return h


def run(h, d, n, out):
    h.process_device(d, n, out)


for rep in range(3):
    h = handle(B); d = h.upload(clip); o = np.zeros(1000, lib.RESULT_DTYPE)
    run(h, d, 2 * B, o[:2 * B])
    t = time.perf_counter(); run(h, d, 1000, o); one = 1000 / (time.perf_counter() - t)
    h.free(d); h.close()
    hh = [handle(B), handle(B)]
    dd = [hh[0].upload(clip[:500]), hh[1].upload(clip[500:])]
    oo = [np.zeros(500, lib.RESULT_DTYPE), np.zeros(500, lib.RESULT_DTYPE)]
    for k in range(2): run(hh[k], dd[k], 2 * B, oo[k][:2 * B])
    th = [threading.Thread(target=run, args=(hh[k], dd[k], 500, oo[k])) for k in range(2)]
    t = time.perf_counter()
    for x in th: x.start()
    for x in th: x.join()
    two = 1000 / (time.perf_counter() - t)
    same = o[:500].tobytes() == oo[0].tobytes() and o[500:].tobytes() == oo[1].tobytes()
    for k in range(2): hh[k].free(dd[k]); hh[k].close()
    print(f"rep {rep}: one handle {one:.1f} frames/s, two concurrent handles {two:.1f} frames/s ({two / one:.3f}x), records identical: {same}", flush=True)

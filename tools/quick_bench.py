"""Developer timing helper (not the judged bench): times the resident-input path for a few batch sizes."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import lib, synth, weights

prec = sys.argv[1] if len(sys.argv) > 1 else "f16"
batches = [int(b) for b in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 8]
hs, ys = weights.make_hrnet_state_dict(0), weights.make_yolo_state_dict("n", 0)
for B in batches:
    h = lib.Handle(batch=B, precision=lib.PRECISIONS[prec], use_graph=int(os.environ.get('GRAPH', '0')))
    t = time.time(); weights.load_into(h, [hs, ys]); print(f"B={B} finalize {time.time()-t:.1f}s", flush=True)
    NB = 6
    frames = synth.clip(0, B * NB, distinct=min(B, 4))
    d = h.upload(frames)
    out = h.process_device(d, B)
    t = time.time(); reps = 3
    for _ in range(reps): out = h.process_device(d, B)
    dt = (time.time() - t) / reps
    print(f"B={B} {prec}: single-batch calls {dt*1e3:.2f} ms/batch  {B/dt:.1f} fps  lib_total_ms={h.timings().total_ms:.2f}", flush=True)
    out = h.process_device(d, B * NB)
    t = time.time()
    out = h.process_device(d, B * NB)
    dt = (time.time() - t)
    print(f"B={B} {prec}: pipelined {NB} batches/call {dt*1e3/NB:.2f} ms/batch  {B*NB/dt:.1f} fps", flush=True)
    h.set_profiling(1)
    out = h.process_device(d, B)
    tm = h.timings()
    print(f"   profiled: total {tm.total_ms:.2f} ms conv {tm.conv_ms:.2f} ms  convs {tm.n_conv_launches} launches {tm.n_launches} "
          f"conv TFLOP/s {tm.conv_flop/ max(tm.conv_ms,1e-9)/1e9:.1f}  n_det {out['n_det'][:3]} H_valid {out['H_valid'][:3]}", flush=True)
    h.free(d); h.close()

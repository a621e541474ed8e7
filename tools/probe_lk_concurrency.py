"""Developer probe: is eagle_clip_flow (K12 + filter) reproducible while ANOTHER handle keeps the GPU busy with the stateless path?
Two handles, two threads (ctypes releases the GIL).  One process runs a list of experiments, each = (co-runner, K12 mode):

    python tools/probe_lk_concurrency.py [exp ...]        exp = <corunner>[:<lk-mode>]
      corunner: idle | f16 | f32 | f16v0 (EAGLE_CONV_FORCE=16,1,0: plain register-staged kernels, no LDS-DMA, no s_setprio)
                | f16nt4 (EAGLE_CONV_FORCE=16,4,0: s_setprio, no LDS-DMA) | ransac | copies
      lk-mode : default | w1 (one wave per key-point) | excl (K12 asks for 150 KB of LDS: alone on its CU)
                | guard (LDS guard words) | trace (per-iteration trace, first divergence printed)
                | verify (end of level: first load vs second load vs LDS, Scharr recomputed) | nt (gray loads bypass the CU's L1)
      corunner also: h_lds h_oob h_mfma h_l1 h_churn h_ldsbyte = synthetic one-feature kernels of tools/lkprobe/hammer.hip
"""
import os
os.environ.setdefault("EAGLE_ENABLE_DEBUG", "1")      # eagle_debug is inert without it
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, "."); sys.path.insert(0, "tests")
import flow_cases
from eagle_amd import lib, synth, weights
from eagle_amd.coordinate_model import CoordinateModel

REPS = int(os.environ.get("PROBE_REPS", "40"))
exps = sys.argv[1:] or ["idle", "f16", "f16:w1", "f16:excl", "f16:guard", "f16:trace", "f32", "f16v0", "f16nt4"]
hs, ys = weights.make_hrnet_state_dict(0), weights.make_yolo_state_dict("n", 0)
frames = np.stack(flow_cases.frames_of("fps25")[:8])
B = CoordinateModel(precision="f16", batch=2, hrnet_state_dict=hs, detector_state_dict=ys)
d = B.handle.upload(frames)
B.handle.clip_open(d, len(frames))
vis = flow_cases.synth.visible_landmarks(2, 60)
kps = np.zeros(len(vis), lib.FLOWKP_DTYPE)
for k, (i, (x, y)) in enumerate(sorted(vis.items())):
    kps[k] = (i, x, y, 0.9)
busy_frames = synth.clip(0, 8)
TR_SHAPE = (57, 3, 12, 8)


def set_mode(m):
    lib.debug("lk_threads", 64 if m.startswith("w1") else 256)
    if m.startswith("w1") and len(m) > 2:
        m = m[2:]
    lib.debug("lk_excl_lds", 150000 if m == "excl" else 0)
    lib.debug("lk_dbg", {"guard": 2, "trace": 1, "verify": 4, "nt": 8}.get(m, 0))


def flow(a):
    return B.handle.clip_flow(a, a + 1, a + 1, kps, raw=True)


def trace():
    return lib.debug("lk_trace", 0, np.zeros(TR_SHAPE, np.int64))


corunners = {}
co_ref = {}
HAMMERS = {"h_lds": 0, "h_oob": 1, "h_mfma": 2, "h_l1": 3, "h_churn": 4, "h_ldsbyte": 5}
_hl = None


def hammer(kind):
    global _hl
    if _hl is None:
        import ctypes
        _hl = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "lkprobe", "libhammer.so"))
    rc = _hl.hammer_launch(kind, 4)
    assert rc == 0, rc


C_ = lib.Handle(device=0)          # weight-less handle: uploads only


_rng = np.random.default_rng(5)
_x48 = _rng.standard_normal((2, 135, 240, 48)).astype(np.float32); _w48 = (_rng.standard_normal((3, 3, 48, 48)) * 0.05).astype(np.float32)
_x16 = _rng.standard_normal((2, 135, 240, 16)).astype(np.float32); _w16 = (_rng.standard_normal((3, 3, 16, 16)) * 0.05).astype(np.float32)
_fr = synth.clip(0, 2)


def _force(v, fn):
    def run():
        if v:
            os.environ["EAGLE_CONV_FORCE"] = v
        try:
            fn()
        finally:
            os.environ.pop("EAGLE_CONV_FORCE", None)
    return run


OPS = {   # single operators of the library as co-runners (each call: upload, ONE kernel, download)
    "op_conv48": lambda: lib.op_conv2d(_x48, _w48, np.zeros(48, np.float32), post=1, precision=lib.PREC_F16),
    "op_conv48v0": _force("16,1,0", lambda: lib.op_conv2d(_x48, _w48, np.zeros(48, np.float32), post=1, precision=lib.PREC_F16)),
    "op_conv16v0": _force("16,1,0", lambda: lib.op_conv2d(_x16, _w16, np.zeros(16, np.float32), post=1, precision=lib.PREC_F16)),
    "op_conv48f32": lambda: lib.op_conv2d(_x48, _w48, np.zeros(48, np.float32), post=1, precision=lib.PREC_F32),
    "op_fuse": lambda: lib.op_fuse_sum(_x48, [_x48[:, ::2, ::2].copy()], precision=lib.PREC_F16),
    "op_pre": lambda: lib.op_preprocess(_fr, precision=lib.PREC_F16),
}


def corunner(name):
    if name in corunners or name in ("idle", "ransac", "copies") or name in HAMMERS or name in OPS:
        return corunners.get(name)
    force = {"f16v0": "16,1,0", "f16nt4": "16,4,0"}.get(name)
    if force:
        os.environ["EAGLE_CONV_FORCE"] = force
    corunners[name] = CoordinateModel(precision="f32" if name == "f32" else "f16", batch=8, hrnet_state_dict=hs, detector_state_dict=ys)
    os.environ.pop("EAGLE_CONV_FORCE", None)
    return corunners[name]


set_mode("default")
ref = [flow(a) for a in range(7)]
set_mode("w1")
w1 = [flow(a) for a in range(7)]
print("one-wave K12 == four-wave K12 on an idle GPU:", all(np.array_equal(x[1].view(np.uint32), y[1].view(np.uint32)) and np.array_equal(x[2], y[2]) for x, y in zip(ref, w1)))
set_mode("trace")
ref_tr = []
for a in range(7):
    flow(a); ref_tr.append(trace())
set_mode("default")

for exp in exps:
    co, _, mode = exp.partition(":")
    mode = mode or "default"
    skip = 0
    if "/" in co:                                          # f16v0/<mask>: the co-runner skips parts of its pipeline (eagle_debug "skip")
        co, sk = co.split("/")
        skip = int(sk)
    lib.debug("skip", skip)
    A = corunner(co)
    stop = False

    def busy():
        pts = np.random.default_rng(0).uniform(0, 700, (30, 2)).astype(np.float32)
        wld = np.random.default_rng(1).uniform(0, 68, (30, 2)).astype(np.float32)
        while not stop:
            if A is not None:
                r_ = A.process_records(busy_frames)
                if skip == 0:
                    if id(A) not in co_ref:
                        co_ref[id(A)] = r_.copy()
                    co_stats[0] += 1
                    co_stats[1] += int(any(r_[f_].tobytes() != co_ref[id(A)][f_].tobytes() for f_ in r_.dtype.names))
            elif co in OPS:
                OPS[co]()
            elif co in HAMMERS:
                hammer(HAMMERS[co])
            elif co == "ransac":
                lib.op_find_homography(pts, wld)
            else:
                dd = C_.upload(busy_frames); C_.free(dd)

    set_mode(mode)
    co_stats = [0, 0]
    if A is not None and skip == 0 and id(A) not in co_ref:
        co_ref[id(A)] = A.process_records(busy_frames).copy()      # the co-runner's own records on an idle GPU
    lib.debug("lk_counters_reset")
    t = threading.Thread(target=busy)
    if co != "idle":
        t.start(); time.sleep(0.5)
    bad = stale_hits = stale_total = stale_shown = 0
    t0 = time.time()
    for rep in range(REPS):
        for a in range(7):
            out, nxt, st = flow(a)
            e1 = not np.array_equal(nxt.view(np.uint32), ref[a][1].view(np.uint32)); e2 = not np.array_equal(st, ref[a][2]); e3 = not np.array_equal(out, ref[a][0])
            tr = None
            if e1 or e2 or e3:
                tr = trace() if mode.endswith("trace") else None      # fetched only on a mismatch: the host-side cadence of a good call is the default one
                bad += 1
                w = np.nonzero((nxt != ref[a][1]).any(1))[0]
                for p_ in w:                                   # is a wrong value the (stale) result of ANOTHER pair's call?
                    hit = [b_ for b_ in range(7) if b_ != a and np.array_equal(nxt[p_].view(np.uint32), ref[b_][1][p_].view(np.uint32))]
                    stale_hits += bool(hit); stale_total += 1
                    if hit and stale_shown < 5:
                        stale_shown += 1
                        print(f"  [{exp}] pair {a} point {p_}: value equals the reference of pair(s) {hit}")
                if bad <= 3:
                    print(f"  [{exp}] pair {a}: lk-points {e1} status {e2} filtered {e3}; points {w[:6].tolist()} got {nxt[w[:3]].tolist()} ref {ref[a][1][w[:3]].tolist()}")
                    if tr is not None:
                        xcc = tr[:len(kps), 0, 11, 3] & 0xF
                        print(f"    XCC of every point's workgroup: {xcc.tolist()}")
                        print(f"    wrong points: {w.tolist()} on XCCs {xcc[w].tolist()}; args seen (max_count, eps2 bits, src|dst, exit reason|last j at level 0): {[hex(int(x)) for x in tr[w[0], 0, 11, [0, 1, 5, 6]]]}")
                        hw = tr[:len(kps), :, 11, 4]
                        moved = [int(p_) for p_ in range(len(kps)) if len(set(int(x) for x in hw[p_] if x)) > 1]
                        print(f"    points whose workgroup reports DIFFERENT HW_ID at the ends of its pyramid levels (wave moved = context save/restore): {moved}")
                        print(f"    HW_ID per level of the wrong points: {[[hex(int(x)) for x in hw[p_]] for p_ in w[:6]]}")
                        for p in w[:3]:
                            m_ = tr[p] != ref_tr[a][p]
                            m_[:, 11, 3:5] = False                  # XCC / HW ids differ by construction
                            dif = np.argwhere(m_)
                            if len(dif):
                                lv = sorted(set(int(x) for x in dif[:, 0]), reverse=True)[0]       # levels run 2 -> 0: the first divergence is at the highest level
                                dl = dif[dif[:, 0] == lv]
                                s0 = int(dl[:, 1].min())
                                print(f"    point {p}: first divergence at level {lv} slot {s0} (0 = level header, 1.. = iteration, 11 = trailer), words {sorted(set(int(x) for x in dl[dl[:, 1] == s0][:, 2]))}")
                                print("      got", [hex(int(x) & (2**64 - 1)) for x in tr[p, lv, s0]])
                                print("      ref", [hex(int(x) & (2**64 - 1)) for x in ref_tr[a][p, lv, s0]])
                                print(f"      level {lv} trailer (max_count, eps2, nxt, xcc, hw, src|dst, exit reason|last j): got", [hex(int(x) & (2**64 - 1)) for x in tr[p, lv, 11, :7]])
                                print(f"      level {lv} trailer: ref", [hex(int(x) & (2**64 - 1)) for x in ref_tr[a][p, lv, 11, :7]])
                                if s0 >= 2:
                                    print("      last common iteration: got", [hex(int(x) & (2**64 - 1)) for x in tr[p, lv, s0 - 1]])
    stop = True
    if co != "idle":
        t.join()
    extra = ""
    if mode.endswith("guard"):
        extra = f"; guard-word faults {lib.debug('lk_counters', 0, np.zeros(16, np.int32))[:7].tolist()}"
    if mode.endswith("verify"):
        c = lib.debug('lk_counters', 0, np.zeros(16, np.int32))
        extra = f"; verify: I reload!=first {c[8]}, I lds!=first {c[9]}, J reload!=first {c[10]}, J lds!=first {c[11]}, Scharr lds!=recomputed {c[12]}, I-window registers != re-derived {c[14]}, DPP sum != LDS-atomic sum {c[15]}, levels checked {c[13]}"
    if co_stats[0]:
        extra += f"; co-runner batches whose records differ from its idle-GPU records: {co_stats[1]} of {co_stats[0]}"
    print(f"{exp}: non-reproducible flow calls {bad} of {REPS * 7} ({time.time() - t0:.1f} s); wrong points equal to another pair's reference: {stale_hits} of {stale_total}{extra}", flush=True)
set_mode("default")
after = sum(not np.array_equal(flow(a)[1].view(np.uint32), ref[a][1].view(np.uint32)) for a in range(7))
print("pairs that still differ once the GPU is idle again (persistent corruption of the inputs):", after, "of 7")
B.handle.clip_close(); B.handle.free(d)
for A in corunners.values():
    A.handle.close()
B.handle.close()

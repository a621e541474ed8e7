"""Victim kernels (hammer.hip) next to the real convolution co-runner (f16 stateless path, EAGLE_CONV_FORCE=16,1,0)."""
import ctypes, os, sys, threading, time
import numpy as np
sys.path.insert(0, ".")
from eagle_amd import lib, synth, weights
from eagle_amd.coordinate_model import CoordinateModel
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhammer.so"))
L.victim_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong)]
NAMES = ["dpp wave sum", "32x32->64 integer product", "fp64", "sqrt / rcp / rint / floor", "int64->float + fma", "LDS byte/word reads", "four-wave exact sum (DPP+LDS+barrier)", "ds_bpermute shuffle sum", "plain int/float VALU", "packed fp32 v_pk_mul/add_f32 vs scalar (thread count = in-kernel disagreements)"]
ROUNDS, N = int(os.environ.get("V_ROUNDS", "4000")), int(os.environ.get("V_N", "400"))
bt = ctypes.c_longlong(0)
for k in range(len(NAMES)):
    assert L.victim_run(k, 3, ROUNDS, ctypes.byref(bt)) == 0, f"victim {k} not reproducible on an idle GPU"
if os.environ.get("V_PRIME", "1") == "1":          # the condition under which K12 fails most: the hammer library has run in this process before
    for _ in range(50):
        assert L.hammer_launch(2, 4) == 0
os.environ["EAGLE_CONV_FORCE"] = os.environ.get("V_FORCE", "16,1,0")
A = CoordinateModel(precision="f16", batch=8, hrnet_state_dict=weights.make_hrnet_state_dict(0), detector_state_dict=weights.make_yolo_state_dict("n", 0))
frames = synth.clip(0, 8)
stop = False
lib.debug("skip", int(os.environ.get("V_SKIP", "126")))       # default: HRNet convolutions only (the co-runner under which K12 fails most)
def busy():
    while not stop:
        A.process_records(frames)
t = threading.Thread(target=busy); t.start(); time.sleep(0.5)
for k, nm in list(enumerate(NAMES))[::-1]:
    bad = L.victim_run(k, N, ROUNDS, ctypes.byref(bt))
    print(f"conv co-runner: victim {k} ({nm}): {bad} of {N} launches differ, {bt.value} thread results", flush=True)
stop = True; t.join()
A.handle.close()

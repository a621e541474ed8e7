"""Which class of K12's arithmetic goes wrong next to an MFMA-heavy co-runner?  (developer probe; see hammer.hip)"""
import ctypes, os, sys, threading, time
L = ctypes.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libhammer.so"))
L.victim_run.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(ctypes.c_longlong)]
NAMES = ["dpp wave sum", "32x32->64 integer product", "fp64", "sqrt / rcp / rint / floor", "int64->float + fma", "LDS byte/word reads", "four-wave exact sum (DPP+LDS+barrier)", "ds_bpermute shuffle sum", "plain int/float VALU", "packed fp32 v_pk_mul/add_f32 vs scalar (thread count = in-kernel disagreements)"]
hammers = [int(x) for x in sys.argv[1:]] or [2]
ROUNDS, N = 2000, 300
bt = ctypes.c_longlong(0)
for k in range(len(NAMES)):
    assert L.victim_run(k, 3, ROUNDS, ctypes.byref(bt)) == 0, f"victim {k} not reproducible on an idle GPU"
for hk in hammers:
    stop = False
    def busy():
        while not stop:
            assert L.hammer_launch(hk, 4) == 0
    t = threading.Thread(target=busy); t.start(); time.sleep(0.3)
    for k, nm in enumerate(NAMES):
        bad = L.victim_run(k, N, ROUNDS, ctypes.byref(bt))
        print(f"hammer {hk}: victim {k} ({nm}): {bad} of {N} launches differ, {bt.value} thread results", flush=True)
    stop = True; t.join()

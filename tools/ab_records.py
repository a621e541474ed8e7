"""Are the records of two builds of the library byte-identical?  python tools/ab_records.py <libA.so> <libB.so> [precision] [detector_precision]
Each build runs in its own process (EAGLE_HIP_LIB) over 12 synthetic frames (default handle unless a precision is named) and dumps its records."""
import hashlib
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CODE = r'''
import sys, numpy as np
sys.path.insert(0, %r)
from eagle_amd import lib, synth, weights
kw = {}
if len(sys.argv) > 1: kw["precision"] = lib.PRECISIONS[sys.argv[1]]
if len(sys.argv) > 2: kw["det_precision"] = lib.PRECISIONS[sys.argv[2]] + 1
h = lib.Handle(batch=5, **kw)
weights.load_into(h, [weights.make_hrnet_state_dict(0), weights.make_yolo_state_dict("n", 0)])
frames = np.stack([synth.frame(0, t) for t in range(10)] + [synth.noise_frame(1), synth.frame(2, 5)])
r = h.process(frames)
sys.stdout.buffer.write(r.tobytes())
''' % ROOT
outs = []
for so in sys.argv[1:3]:
    r = subprocess.run([sys.executable, "-c", CODE] + sys.argv[3:], capture_output=True, env=dict(os.environ, EAGLE_HIP_LIB=os.path.abspath(so)))
    if r.returncode:
        print(r.stderr.decode()[-2000:])
        sys.exit(1)
    outs.append(r.stdout)
print("records", len(outs[0]), "bytes each;", "IDENTICAL" if outs[0] == outs[1] else "DIFFERENT", hashlib.md5(outs[0]).hexdigest()[:12], hashlib.md5(outs[1]).hexdigest()[:12])
sys.exit(0 if outs[0] == outs[1] else 2)

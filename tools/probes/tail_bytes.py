"""Developer probe: which stage of the clip session writes the 4 trailing padding bytes of the records?"""
import sys
import numpy as np
sys.path.insert(0, ".")
from eagle_amd import synth
from eagle_amd.coordinate_model import CoordinateModel
frames = np.stack([synth.frame(0, t) for t in range(16)])
m = CoordinateModel(precision="f16", batch=4); h = m.handle
d = h.upload(frames)
def tail(tag):
    r = h.clip_fetch(len(frames)); b = np.frombuffer(r.tobytes(), np.uint8).reshape(len(frames), -1)[:, -4:]
    print(tag, "non-zero tails:", int((b != 0).any(1).sum()), b[:3].tolist())
h.clip_open(d, len(frames)); tail("after open")
h.clip_detect_objects(0, len(frames)); tail("after detect_objects")
h.clip_detect_keypoints(0, 8, 2); tail("after detect_keypoints")
h.clip_run(0, len(frames), 8, 25, False, wait=True); tail("after run")
h.clip_close(); h.free(d)
r = m.process_records(frames); b = np.frombuffer(r.tobytes(), np.uint8).reshape(len(frames), -1)[:, -4:]
print("stateless path non-zero tails:", int((b != 0).any(1).sum()))
h.close()
base = np.stack([synth.frame(0, t) for t in range(40)])
clip = np.ascontiguousarray(np.concatenate([base, base[::-1]] * 13)[:1000])
for n in (16, 64, 400, 1000):
    m = CoordinateModel(precision="f16", batch=50); r = m.flow_records(clip[:n], 8, 25); m.handle.close()
    b = np.frombuffer(r.tobytes(), np.uint8).reshape(n, -1)[:, -4:]
    print("flow_records", n, "frames: non-zero tails", int((b != 0).any(1).sum()), b[:2].tolist(), "as float", np.frombuffer(b[:2].tobytes(), np.float32), "as int", np.frombuffer(b[:2].tobytes(), np.int32))

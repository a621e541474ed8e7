import sys, numpy as np
sys.path.insert(0, "/root/repo")
from eagle_amd import lib, synth, weights
h = lib.Handle(batch=50)
weights.load_into(h, [weights.make_hrnet_state_dict(0), weights.make_yolo_state_dict("n", 0)])
clip = synth.clip(seed=0, n=20); clip = np.concatenate([clip, clip, clip[:10]])
d = h.upload(clip); o = np.zeros(50, lib.RESULT_DTYPE)
h.process_device(d, 50, o)
h.set_profiling(1)
for _ in range(3): h.process_device(d, 50, o)
for name, ms, launches, nbytes, flop in h.kernel_times():
    if not name.startswith("conv "): print(f"{name:20s} {ms / launches * 1e3:9.1f} us x{launches}")
print("H_valid frames:", int(o["H_valid"].sum()), "of 50")

import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from eagle_amd import lib, weights
out = sys.argv[1]; B = int(sys.argv[2]); det = sys.argv[3] if len(sys.argv) > 3 else "n"; imgsz = int(sys.argv[4]) if len(sys.argv) > 4 else 640
fh, fw = (1080, 1920) if det == "l" else (720, 1280)
os.environ["EAGLE_DUMP_LAYERS"] = out
h = lib.Handle(batch=B, det_variant=det, det_imgsz=imgsz, frame_h=fh, frame_w=fw)
weights.load_into(h, [weights.make_hrnet_state_dict(0), weights.make_yolo_state_dict(det, 0)])
h.close()

"""Parity study for DESIGN §10 item 5 (not part of the product, nothing is built on it yet): what would Winograd F(2x2, 3x3) do to the records?
Every 3x3 stride-1 convolution of both networks evaluated as Winograd in fp32 (torch-CPU: 16 batched [Cout x Cin] . [Cin x tiles] products,
input / output transforms in fp32) against the direct fp32 evaluation of the same backend, same weights, same frames — next to the spread
between the oracle's two DIRECT fp32 backends (tests/golden/fp32_order_noise.py), which is the yardstick the f32s tolerances were set from.
CPU only; 5 s per frame.  Usage: python tests/golden/winograd_noise.py [n_frames]

Measured 2026-10-03 (8 vCPU container), five frames of the cfg-2 synthetic clip, 246 Winograd layers per frame:
  heat-map indices 285/285 equal, scores within 3.6e-7, logits within 5.4e-7 of max|logit| (8.8e-7 between the two direct backends),
  detection counts equal on all frames, confidences within 1.5e-6 (2.4e-6), every box within 3.4e-4 px of a box of the other side; index-wise 4 of
  1313 integer boxes and 2 classes differ (one swapped near-tie pair on frame 0, as between the direct backends), H bit-identical on all
  frames, 2 of 412 pitch integers (the swapped pair).
  => in fp32, Winograd F(2x2, 3x3) stays INSIDE the noise between two legitimate direct evaluations on these networks."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

from eagle_amd import synth, weights  # noqa: E402
from oracle import nets, pipeline  # noqa: E402

G = torch.tensor([[1, 0, 0], [.5, .5, .5], [.5, -.5, .5], [0, 0, 1]], dtype=torch.float32)
BT = torch.tensor([[1, 0, -1, 0], [0, 1, 1, 0], [0, -1, 1, 0], [0, 1, 0, -1]], dtype=torch.float32)
AT = torch.tensor([[1, 1, 1, 0], [0, 1, -1, -1]], dtype=torch.float32)


class WinogradBackend(nets.TorchBackend):
    """TorchBackend with every 3x3 stride-1 convolution as Winograd F(2x2, 3x3), all arithmetic fp32"""

    def __init__(self):
        super().__init__()
        self._u = {}
        self.n_wino = 0

    def conv(self, x, wb, stride=1, pre=0, r1=None, r2=None, post=0, f32_out=False):
        w, b = wb
        if w.shape[0] != 3 or stride != 1:
            return super().conv(x, wb, stride, pre, r1, r2, post, f32_out)
        self.n_wino += 1
        key = id(w)
        if key not in self._u:
            g = torch.from_numpy(np.ascontiguousarray(w.transpose(3, 2, 0, 1)))          # [K, C, 3, 3]
            self._u[key] = (torch.einsum("ai,kcij,bj->kcab", G, g, G).contiguous(), torch.from_numpy(b))
        U, bt = self._u[key]
        N, C, H, W = x.shape
        th, tw = (H + 1) // 2, (W + 1) // 2
        xp = torch.zeros((N, C, 2 * th + 2, 2 * tw + 2), dtype=torch.float32)
        xp[:, :, 1:H + 1, 1:W + 1] = x
        d = xp.unfold(2, 4, 2).unfold(3, 4, 2)                                            # [N, C, th, tw, 4, 4]
        V = torch.einsum("ai,nctsij,bj->nctsab", BT, d, BT)
        M = torch.einsum("kcab,nctsab->nktsab", U, V)
        Y = torch.einsum("ia,nktsab,jb->nktisj", AT, M, AT)                               # [N, K, th, 2, tw, 2]
        v = Y.reshape(N, U.shape[0], 2 * th, 2 * tw)[:, :, :H, :W] + bt.view(1, -1, 1, 1)
        v = self._act(v, pre)
        if r1 is not None:
            v = r1 + v
        if r2 is not None:
            v = v + r2
        return self._act(v, post)


def run(backend_obj, hs, ys, frame):
    """one oracle step with the given backend object in place of the torch backend"""
    saved = nets.TorchBackend
    nets.TorchBackend = lambda: backend_obj
    try:
        return pipeline.OracleModel(hs, ys, backend="torch").step(frame, 0)
    finally:
        nets.TorchBackend = saved


def main():
    n_frames = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    torch.set_num_threads(max(1, (os.cpu_count() or 8)))
    hs, ys = weights.make_hrnet_state_dict(0), weights.make_yolo_state_dict("n", 0)
    tot = dict(frames=0, hm_idx=0, n_det=0, dets=0, int_box=0, cls=0, H_valid=0, pitch_int=0, pitch_n=0)
    worst = dict(score=0.0, logit=0.0, conf=0.0, box=0.0, H=0.0)
    for k in range(n_frames):
        f = synth.frame(0, 3 * k)
        t = time.time()
        rec_d, a = run(nets.TorchBackend(), hs, ys, f)
        wb = WinogradBackend()
        rec_w, b = run(wb, hs, ys, f)
        tot["frames"] += 1
        tot["hm_idx"] += int((a["hm_idx"] != b["hm_idx"]).sum())
        worst["score"] = max(worst["score"], float(np.abs(a["hm_score"] - b["hm_score"]).max()))
        worst["logit"] = max(worst["logit"], float(np.abs(a["logits"] - b["logits"]).max() / np.abs(a["logits"]).max()))
        if len(a["dets"]) != len(b["dets"]):
            tot["n_det"] += 1
        n = min(len(a["dets"]), len(b["dets"]))
        tot["dets"] += n
        worst["conf"] = max(worst["conf"], float(np.abs(a["dets"][:n, 4] - b["dets"][:n, 4]).max(initial=0)))
        tot["int_box"] += int((a["dets"][:n, :4].astype(np.int64) != b["dets"][:n, :4].astype(np.int64)).any(1).sum())
        tot["cls"] += int((a["dets"][:n, 5] != b["dets"][:n, 5]).sum())
        # a box differing by index may be a swapped near-tie: distance to the nearest box of the other side
        if n:
            worst["box"] = max(worst["box"], float(max(np.abs(b["dets"][:, :4] - d[:4]).max(1).min() for d in a["dets"])))
        hv_a, hv_b = a["H"] is not None, b["H"] is not None
        tot["H_valid"] += int(hv_a != hv_b)
        if hv_a and hv_b:
            Ha, Hb = np.asarray(a["H"], np.float64), np.asarray(b["H"], np.float64)
            worst["H"] = max(worst["H"], float(np.abs(Ha - Hb).max() / np.abs(Ha).max()))
        for cname, objs in rec_d["Coordinates"].items():
            for oid, o in objs.items():
                o2 = rec_w["Coordinates"].get(cname, {}).get(oid)
                tot["pitch_n"] += 1
                tot["pitch_int"] += int(o2 is None or o2.get("Transformed_Coordinates") != o.get("Transformed_Coordinates"))
        print(f"frame {k} ({time.time() - t:.0f} s, {wb.n_wino} Winograd layers): {tot} {worst}", flush=True)
    print("TOTAL", tot, worst)


if __name__ == "__main__":
    main()

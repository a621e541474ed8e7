"""Two (or more) REAL ranks on ONE GPU, records compared with a single-rank run (developer / -m gpu test; no 8-GPU node needed).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P tools/shard_check.py

Every rank opens its own library handle on HIP device 0 (gloo process group: RCCL cannot put two ranks on one device), runs
  (a) its contiguous chunk of ONE frame-sharded clip through the stateless path (BASELINE configs[3] shape), and
  (b) its own clip(s) of a clip-sharded batch through the reference cadence (configs[4] shape: the flow cadence is sequential in a clip),
gathers the fixed-size records (eagle_amd.shard, transport "dist"), and rank 0 checks them field by field against the records the
same handle produces for the whole workload on its own.  Prints SHARD_CHECK_OK."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def same(a, b):
    return all(a[f].tobytes() == b[f].tobytes() for f in a.dtype.names)


def main():
    import torch.distributed as dist
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    from eagle_amd import clip as clipmod
    from eagle_amd import lib, shard, synth, weights
    hs, ys = weights.make_hrnet_state_dict(0), weights.make_yolo_state_dict("n", 0)
    B = int(os.environ.get("SHARD_CHECK_BATCH", "4"))
    h = lib.Handle(device=0, batch=B, precision=lib.PRECISIONS[os.environ.get("SHARD_CHECK_PRECISION", "f32s")])
    weights.load_into(h, [hs, ys])
    # (a) frame-sharded: ragged on purpose (default world * B + 3 frames; SHARD_CHECK_FRAMES overrides, e.g. 25 frames on 8 ranks = chunks of 4:
    #     rank 6 holds one frame, rank 7 none)
    n = int(os.environ.get("SHARD_CHECK_FRAMES", str(world * B + 3)))
    frames = synth.clip(seed=5, n=n)
    lo, hi = shard.shard_range(n, rank, world)
    local = h.process(frames[lo:hi]) if hi > lo else np.zeros(0, lib.RESULT_DTYPE)
    allr = shard.gather_records(local, n, rank, world, transport="dist")
    # (b) clip-sharded cadence: world + 1 clips of different lengths, keypoint_interval 3, homography_interval 6
    lengths = [7 + 2 * (k % 3) for k in range(world + 1)]
    clips = [synth.clip(seed=20 + k, n=L) for k, L in enumerate(lengths)]
    a, b = shard.shard_range(len(clips), rank, world)
    mine = []
    for k in range(a, b):
        d = h.upload(clips[k])
        mine.append(clipmod.run_clip(h, d, lengths[k], 3, 6))
        h.free(d)
    allc = shard.gather_clip_records(mine, lengths, rank, world, transport="dist")
    if rank == 0:
        whole = h.process(frames)
        assert len(allr) == n and same(allr, whole), "frame-sharded records differ from the single-rank run"
        for k, c in enumerate(clips):
            d = h.upload(c)
            ref = clipmod.run_clip(h, d, lengths[k], 3, 6)
            h.free(d)
            assert len(allc[k]) == lengths[k] and same(allc[k], ref), f"clip-sharded records of clip {k} differ from the single-rank run"
        print(f"SHARD_CHECK_OK world={world} frames={n} clips={lengths}", flush=True)
    h.close()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()

"""Frame sharding across the GPUs of one node: one process per GPU, rank r owns the contiguous chunk
``[r*ceil(F/R), min(F, (r+1)*ceil(F/R)))``, weights replicated, no communication during the clip, and ONE collective
at the end: an all-gather of the fixed-size per-frame records.

Two transports for that one collective:
  * ``rccl``  — the library's own ``eagle_gather`` (ncclAllGather on the handle's stream; RCCL over xGMI);
  * ``dist``  — ``torch.distributed.all_gather`` on the default process group (gloo in the CPU tests, RCCL when the
               group was created with backend "nccl").
The reference has no multi-process path at all (single process, single device: coordinate_model.py:23-29)."""
import numpy as np

from .lib import RESULT_DTYPE


def chunk_size(n_frames, world):
    return -(-n_frames // world)


def shard_range(n_frames, rank, world):
    c = chunk_size(n_frames, world)
    return min(n_frames, rank * c), min(n_frames, (rank + 1) * c)


def _pad(local, c):
    if len(local) == c:
        return np.ascontiguousarray(local)
    out = np.zeros(c, local.dtype)
    out[: len(local)] = local
    return out


def gather_records(local, n_frames, rank, world, handle=None, transport="rccl", out=None, force=False):
    """local: this rank's records (structured array).  Returns all ``n_frames`` records in frame order on every rank.
    ``out``: optional preallocated (and pre-touched) [chunk_size*world] result buffer, used when no rank is ragged.
    ``force``: run the collective even at world 1 (bench.py --force-multirank-path: the one-GPU test of the N > 1 code path)."""
    if world == 1 and not force:
        return np.ascontiguousarray(local)
    c = chunk_size(n_frames, world)
    padded = _pad(local, c)
    if transport == "rccl":
        if handle is None:
            raise ValueError("transport='rccl' needs the library handle (eagle_comm_init must have been called)")
        allr = handle.gather(padded, world, out if (out is not None and len(out) == c * world) else None)
    elif transport == "dist":
        from . import lib
        lib.require_torch_first()
        import torch
        import torch.distributed as dist
        raw = torch.from_numpy(padded.view(np.uint8).reshape(-1).copy())
        if dist.get_backend() == "nccl":
            raw = raw.cuda()
        outs = [torch.empty_like(raw) for _ in range(world)]
        dist.all_gather(outs, raw)
        allr = np.concatenate([o.cpu().numpy() for o in outs]).view(local.dtype)
    else:
        raise ValueError(transport)
    if c * world == n_frames:          # no ragged rank: the gathered buffer already is the clip, in frame order
        return allr
    keep = []
    for r in range(world):
        lo, hi = shard_range(n_frames, r, world)
        keep.append(allr[r * c: r * c + (hi - lo)])
    return np.concatenate(keep)


def init_rccl(handle, rank, world):
    """Bootstrap the library's RCCL communicator: rank 0 creates the ncclUniqueId, torch.distributed broadcasts it."""
    from . import lib
    lib.require_torch_first()
    import torch
    import torch.distributed as dist
    uid = torch.zeros(128, dtype=torch.uint8)
    if rank == 0:
        uid = torch.frombuffer(bytearray(lib.comm_unique_id()), dtype=torch.uint8).clone()
    if dist.get_backend() == "nccl":
        uid = uid.cuda()
    dist.broadcast(uid, 0)
    handle.comm_init(rank, world, bytes(uid.cpu().numpy().tobytes()))


# ---- stateful cadences (optical-flow key-point propagation, eagle_amd/clip.py): shard by CLIP ---------------------------------
# With keypoint_interval > 1 frame i depends on frame i-1 (cm.py:207-213: prev_keypoints, prev_gray, the homography and its retry
# flag), so a clip is the unit of work (BASELINE.json configs[4]: 8 clips on 8 GPUs).  Rank r owns the contiguous range of clips
# shard_range(n_clips, r, world); the one collective is again an all-gather of fixed-size records, here of ragged per-rank totals.
def gather_clip_records(local_clips, clip_lengths, rank, world, handle=None, transport="rccl"):
    """local_clips: this rank's clips' records (list of structured arrays, in clip order); clip_lengths: frames of EVERY clip.
    Returns the records of all clips (list, clip order) on every rank."""
    n_clips = len(clip_lengths)
    lo, hi = shard_range(n_clips, rank, world)
    assert len(local_clips) == hi - lo and [len(c) for c in local_clips] == list(clip_lengths[lo:hi])
    if world == 1:
        return [np.ascontiguousarray(c) for c in local_clips]
    totals = [sum(clip_lengths[slice(*shard_range(n_clips, r, world))]) for r in range(world)]
    c = max(max(totals), 1)
    local = np.concatenate(local_clips) if local_clips else np.zeros(0, RESULT_DTYPE)
    allr = gather_records(_pad(local, c), c * world, rank, world, handle=handle, transport=transport)
    out = []
    for r in range(world):
        a, b = shard_range(n_clips, r, world)
        off = r * c
        for k in range(a, b):
            out.append(allr[off: off + clip_lengths[k]])
            off += clip_lengths[k]
    return out

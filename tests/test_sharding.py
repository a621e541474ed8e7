"""CPU tests of the N>1 path: contiguous frame sharding + the single end-of-clip gather of fixed-size records,
run as two real processes over gloo (the GPU box runs the same code over RCCL)."""
import os
import socket
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r'''
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from eagle_amd import shard
from eagle_amd.lib import RESULT_DTYPE
n_frames = int(sys.argv[2])
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lo, hi = shard.shard_range(n_frames, rank, world)
local = np.zeros(hi - lo, RESULT_DTYPE)
for k, f in enumerate(range(lo, hi)):                 # a record that identifies its frame
    local[k]["n_det"] = f
    local[k]["n_kp"] = 1000 + f
    local[k]["H"][:] = np.arange(9) + f
    local[k]["det"][0]["conf"] = f / 7.0
    local[k]["hm_idx"][:] = f
allr = shard.gather_records(local, n_frames, rank, world, transport="dist")
assert len(allr) == n_frames, len(allr)
assert allr["n_det"].tolist() == list(range(n_frames))
assert allr["n_kp"].tolist() == [1000 + f for f in range(n_frames)]
assert all(np.array_equal(allr[f]["H"], np.arange(9) + f) for f in range(n_frames))
assert np.allclose(allr["det"]["conf"][:, 0], np.arange(n_frames) / 7.0)
dist.barrier()
dist.destroy_process_group()
print("rank", rank, "ok", lo, hi)
'''


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(world, n_frames, tmp_path):
    w = tmp_path / "worker.py"
    w.write_text(WORKER)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(w), ROOT, str(n_frames)], env=env, stdout=subprocess.PIPE,
                                      stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_two_rank_gather_even(tmp_path):
    _run(2, 10, tmp_path)


def test_two_rank_gather_ragged(tmp_path):
    _run(2, 7, tmp_path)            # chunks of 4 and 3: the padded slot must be dropped


def test_eight_rank_gather_ragged_with_an_empty_rank(tmp_path):
    """The driver's N = 8 shape on CPU: 25 frames over 8 ranks = chunks of 4 — ranks 0..5 full, rank 6 one frame, rank 7 NO frame."""
    from eagle_amd import shard
    assert [shard.shard_range(25, r, 8) for r in (5, 6, 7)] == [(20, 24), (24, 25), (25, 25)]
    _run(8, 25, tmp_path)


def test_eight_rank_gather_fewer_frames_than_ranks(tmp_path):
    _run(8, 3, tmp_path)            # chunks of 1: five ranks hold nothing


def test_shard_ranges_cover_clip():
    from eagle_amd import shard
    for n in (0, 1, 7, 8, 1000, 8001):
        for w in (1, 2, 3, 8):
            r = [shard.shard_range(n, k, w) for k in range(w)]
            assert r[0][0] == 0 and r[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert all(hi - lo <= shard.chunk_size(n, w) for lo, hi in r)


def test_world_one_gather_is_identity():
    from eagle_amd import shard
    from eagle_amd.lib import RESULT_DTYPE
    a = np.zeros(3, RESULT_DTYPE)
    a["n_det"] = [1, 2, 3]
    assert shard.gather_records(a, 3, 0, 1)["n_det"].tolist() == [1, 2, 3]


CLIP_WORKER = r'''
import os, sys
import numpy as np
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from eagle_amd import shard
from eagle_amd.lib import RESULT_DTYPE
lengths = [int(v) for v in sys.argv[2].split(",")]
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
lo, hi = shard.shard_range(len(lengths), rank, world)
local = []
for c in range(lo, hi):                                # a record that identifies its clip and frame
    a = np.zeros(lengths[c], RESULT_DTYPE)
    a["n_det"] = c
    a["n_kp"] = np.arange(lengths[c])
    local.append(a)
allc = shard.gather_clip_records(local, lengths, rank, world, transport="dist")
assert [len(a) for a in allc] == lengths
for c, a in enumerate(allc):
    assert (a["n_det"] == c).all() and a["n_kp"].tolist() == list(range(lengths[c]))
dist.barrier()
dist.destroy_process_group()
'''


def _run_clips(world, lengths, tmp_path):
    w = tmp_path / "clip_worker.py"
    w.write_text(CLIP_WORKER)
    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE=str(world), LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(w), ROOT, lengths], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=180)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o


def test_two_rank_clip_sharding_for_stateful_cadences(tmp_path):
    """configs[4] shape: whole clips per rank (the optical-flow cadence is sequential within a clip), ragged totals."""
    _run_clips(2, "5,3,9,1,4", tmp_path)


def test_eight_rank_clip_sharding_nine_clips(tmp_path):
    """configs[4] on the driver's node shape: 9 clips for 8 ranks (chunks of 2 clips: ranks 0..3 two clips, rank 4 one, ranks 5..7 none)."""
    _run_clips(8, "5,3,9,1,4,7,2,6,8", tmp_path)

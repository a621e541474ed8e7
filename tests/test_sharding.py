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


# ---- CPU placement of ranks (round 5; VERDICT r4 task 1c): eagle_amd/shard.py::plan_rank_cpus / bind_rank_cpus ---------------------------------------
def _fake_sysfs(root, gpu_nodes, node_cpus, cpu_nodes=2):
    """A sysfs tree with `cpu_nodes` CPU-only KFD nodes followed by one KFD GPU node per entry of gpu_nodes (its NUMA node, or -1)."""
    nodes = root / "class" / "kfd" / "kfd" / "topology" / "nodes"
    for i in range(cpu_nodes):
        d = nodes / str(i); d.mkdir(parents=True)
        (d / "properties").write_text("cpu_cores_count 64\nsimd_count 0\ndrm_render_minor 0\n")
    for g, nn in enumerate(gpu_nodes):
        d = nodes / str(cpu_nodes + g); d.mkdir(parents=True)
        (d / "properties").write_text(f"cpu_cores_count 0\nsimd_count 1024\ndrm_render_minor {128 + g}\n")
        dev = root / "class" / "drm" / f"renderD{128 + g}" / "device"; dev.mkdir(parents=True)
        (dev / "numa_node").write_text(f"{nn}\n")
    for n, cl in node_cpus.items():
        d = root / "devices" / "system" / "node" / f"node{n}"; d.mkdir(parents=True)
        (d / "cpulist").write_text(cl + "\n")
    return str(root)


def test_rank_cpu_plan_from_sysfs(tmp_path):
    from eagle_amd import shard
    assert shard.parse_cpulist("0-3,8,10-11") == {0, 1, 2, 3, 8, 10, 11} and shard.parse_cpulist("") == set()
    # a two-socket MI355X host: GPUs 0-3 on node 0 (CPUs 0-63,128-191), GPUs 4-7 on node 1
    sysfs = _fake_sysfs(tmp_path / "a", [0, 0, 0, 0, 1, 1, 1, 1], {0: "0-63,128-191", 1: "64-127,192-255"})
    assert shard.visible_gpu_numa_nodes(sysfs, env={}) == [0, 0, 0, 0, 1, 1, 1, 1]
    allowed = set(range(256))
    plans = [shard.plan_rank_cpus(r, 8, allowed, sysfs, env={}) for r in range(8)]
    sets = [set(p[0]) for p in plans]
    assert all(len(s) == 32 for s in sets) and all(not (sets[i] & sets[j]) for i in range(8) for j in range(i))      # disjoint 32-CPU slices
    node0, node1 = shard.parse_cpulist("0-63,128-191"), shard.parse_cpulist("64-127,192-255")
    assert all(sets[r] <= node0 for r in range(4)) and all(sets[r] <= node1 for r in range(4, 8)) and "slice 2/4" in plans[1][1]
    # one rank alone: the whole node of ITS GPU, restricted to the cgroup's CPUs
    assert set(shard.plan_rank_cpus(0, 1, set(range(0, 200)), sysfs, env={})[0]) == node0 & set(range(200))
    # ROCR_VISIBLE_DEVICES re-maps device 0 to the physical GPU 5 (node 1)
    assert set(shard.plan_rank_cpus(0, 1, allowed, sysfs, env={"ROCR_VISIBLE_DEVICES": "5"})[0]) == node1
    # eight ranks sharing device 0 (bench.py --shared-gpu): all on node 0, sliced eight ways
    p = [shard.plan_rank_cpus(r, 8, allowed, sysfs, env={}, rank_devs=[0] * 8) for r in range(8)]
    assert all(set(c) <= node0 and len(c) == 16 for c, _ in p) and len(set().union(*[set(c) for c, _ in p])) == 128
    # no NUMA information (numa_node -1): even split when there are enough CPUs, otherwise the mask is left alone
    sysfs2 = _fake_sysfs(tmp_path / "b", [-1] * 8, {})
    assert shard.plan_rank_cpus(3, 8, set(range(64)), sysfs2, env={}) == (list(range(24, 32)), "no NUMA information: even split 4/8")
    assert shard.plan_rank_cpus(3, 8, set(range(8)), sysfs2, env={}) == (list(range(8)), "no NUMA information: unchanged")
    assert shard.plan_rank_cpus(0, 1, {2, 5}, str(tmp_path / "missing"), env={}) == ([2, 5], "no NUMA information: unchanged")


BIND_WORKER = r'''
import os, sys, json
import torch.distributed as dist
sys.path.insert(0, sys.argv[1])
from eagle_amd import shard
dist.init_process_group("gloo")
rank, world = dist.get_rank(), dist.get_world_size()
rep = shard.bind_rank_cpus(rank, world, sysfs=sys.argv[2], env={}, min_slice=1)
import threading
seen = []
t = threading.Thread(target=lambda: seen.append(sorted(os.sched_getaffinity(0))))      # a thread created AFTER the call (the copy workers) inherits the mask
t.start(); t.join()
out = [None] * world
dist.all_gather_object(out, (rank, sorted(os.sched_getaffinity(0)), seen[0], rep))
dist.barrier(); dist.destroy_process_group()
if rank == 0:
    print("MASKS " + json.dumps(out))
'''


def test_rank_cpu_mask_is_applied_at_world_8(tmp_path):
    """Eight gloo ranks, a fake two-node topology over the CPUs this container really has: every rank ends up on its GPU's node (and slice), and a
    thread it starts afterwards inherits the mask."""
    import json
    have = sorted(os.sched_getaffinity(0))
    if len(have) < 2:
        import pytest
        pytest.skip("one usable CPU")
    half = len(have) // 2
    def cl(c): return ",".join(str(x) for x in c)
    sysfs = _fake_sysfs(tmp_path / "s", [0, 0, 0, 0, 1, 1, 1, 1], {0: cl(have[:half]), 1: cl(have[half:])})
    w = tmp_path / "bind_worker.py"
    w.write_text(BIND_WORKER)
    port = _free_port()
    procs = []
    for r in range(8):
        env = dict(os.environ, RANK=str(r), WORLD_SIZE="8", LOCAL_RANK=str(r), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="1")
        procs.append(subprocess.Popen([sys.executable, str(w), ROOT, sysfs], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    line = [ln for ln in outs[0].splitlines() if ln.startswith("MASKS ")][0]
    for rank, mask, thread_mask, rep in json.loads(line[6:]):
        node = have[:half] if rank < 4 else have[half:]
        k = rank % 4
        want = node[len(node) * k // 4: len(node) * (k + 1) // 4] if len(node) >= 4 else node
        assert mask == want and thread_mask == want, (rank, mask, want, rep)
        assert rep["cpus"] == len(want) and ("numa node" in rep["reason"])

"""CPU: `python bench.py --gpus N` against the launcher's environment (VERDICT r5 task 3).  Without a launcher and N > 1 the process becomes the launcher
(fresh child ranks through torch.distributed.run, the driver's own command shape); with a launcher that set another WORLD_SIZE it refuses.  Until round 5
`python bench.py --gpus 8` alone ran rank 0 on one GPU and printed "n_gpus": 1."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "LOCAL_WORLD_SIZE")}
    e.update(OMP_NUM_THREADS="2", **kw)
    return e


def test_a_world_size_that_is_not_gpus_is_refused_before_anything_runs():
    for ws, gpus in (("1", "8"), ("4", "2"), ("2", "1")):
        r = subprocess.run([sys.executable, BENCH, "--gpus", gpus, "--steps", "1"], capture_output=True, text=True, timeout=120, cwd=ROOT, env=_env(WORLD_SIZE=ws, RANK="0"))
        assert r.returncode != 0 and r.stdout.strip() == "", (ws, gpus, r.stdout[-500:])
        assert f"--gpus {gpus}" in r.stderr and f"WORLD_SIZE={ws}" in r.stderr, r.stderr[-500:]


def test_gpus_n_without_a_launcher_builds_the_drivers_command():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "8", "--steps", "5", "--warmup", "2", "--print-launch"], capture_output=True, text=True, timeout=120, cwd=ROOT, env=_env())
    assert r.returncode == 0, r.stderr[-1000:]
    cmd = json.loads(r.stdout.strip().splitlines()[-1])["launch"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=8" in cmd
    i = cmd.index("--master-addr")
    assert cmd[i + 1] == "127.0.0.1" and cmd[i + 2] == "--master-port" and 1024 < int(cmd[i + 3]) < 65536
    j = cmd.index(BENCH)
    assert cmd[j + 1:] == ["--gpus", "8", "--steps", "5", "--warmup", "2"]          # the parent's own arguments, minus --print-launch


def test_self_launch_really_starts_the_ranks_and_relays_their_failure():
    """No GPU in this container: the two child ranks rendezvous over gloo, fail to create a handle, and the parent must hand that on — a non-zero exit code and NO
    result line (the success path is tests/test_gpu_edges.py::test_bench_self_launch_reports_the_gpu_count_it_was_asked_for, on a GPU)."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "1", "--warmup", "0", "--batch", "1", "--distinct", "1", "--shared-gpu", "--backend", "gloo",
                        "--no-cpu-baseline", "--no-extras"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=_env(HIP_VISIBLE_DEVICES="-1"))
    assert "starting 2 ranks" in r.stderr, r.stderr[-2000:]
    assert r.returncode != 0, (r.stdout[-500:], r.stderr[-2000:])
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")], r.stdout[-500:]

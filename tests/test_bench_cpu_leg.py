"""CPU: the bench's cpu_baseline leg runs in a child process (`bench.py --cpu-baseline-only`) so that the process that touches the GPU never
imports torch at N = 1.  Here: the child's contract — one JSON object on the last stdout line with the fields the measurement
contract names — on a one-frame sample."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpu_baseline_child_prints_the_contract_object():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--cpu-baseline-only", "--cpu-frames", "1", "--distinct", "1", "--batch", "1"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT, env=dict(os.environ, OMP_NUM_THREADS="4"))
    assert r.returncode == 0, r.stderr[-2000:]
    d = json.loads(r.stdout.strip().splitlines()[-1])
    assert set(d) >= {"value", "unit", "cores", "kind", "sample"} and d["kind"] == "port" and d["unit"] == "frames/s"
    assert d["value"] > 0 and d["cores"] >= 1 and "oracle/pipeline.py" in d["sample"]


def test_bench_main_process_does_not_import_torch_at_one_gpu():
    """Static check of the import discipline: torch appears in bench.py only inside cpu_baseline() (the child) and behind `if multi:`."""
    src = open(os.path.join(ROOT, "bench.py")).read()
    lines = [ln for ln in src.splitlines() if ln.strip().startswith("import torch")]
    assert len(lines) == 3, lines                                     # cpu_baseline(): 1; the multi-rank branch: torch + torch.distributed
    main = src[src.index("def main():"):]
    assert main.index("if multi:") < main.index("import torch") < main.index("from eagle_amd import lib")      # torch before the HIP library

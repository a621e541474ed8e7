"""The shipped code objects are free of the wide-store data hazard (eagle_amd/csrc/bneck.hip, phase 3): no VALU write to a >64-bit store's data registers within
two issue slots behind it.  hipcc's hazard recognizer misses the SGPR-soffset and inline-asm forms; both have produced wrong bytes on gfx950 (rounds 6)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.exists("/opt/rocm/lib/llvm/bin/llvm-objdump"), reason="no llvm-objdump")
def test_built_library_has_no_store_data_hazard():
    so = os.path.join(ROOT, "eagle_amd", "libeagle_hip.so")
    if not os.path.exists(so):
        import __graft_entry__
        __graft_entry__.build()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "isa_store_hazard.py")], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert " 0 findings" in r.stdout

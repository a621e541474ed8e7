"""CPU test: regression guard on the gfx950 machine code of the kernels with a bit-exactness claim.

DESIGN.md §8c: with hipcc 7.2's SLP vectoriser on, K12 (`lk_kernel`) was built with packed fp32 instructions
(v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) and returned different sub-pixel results whenever f16-MFMA-heavy waves
shared its SIMD; the library is therefore built with -fno-slp-vectorize.  Nothing else stops a toolchain bump, a loop
vectoriser or a float2 idiom from re-introducing those instructions, so this test carves the gfx950 code objects out
of eagle_amd/libeagle_hip.so (clang offload bundles in .hip_fatbin), disassembles the guarded kernels and asserts that
none of them contains a packed fp32 arithmetic instruction."""
import os
import re
import struct
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "eagle_amd", "libeagle_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
# every kernel whose results are compared bit for bit with the oracle (or with the reference's own loop) in fp32 arithmetic
GUARDED = ["lk_kernel", "chain_kernel", "post_kernel", "team_color_kernel", "flow_filter_kernel", "nms_kernel", "yolo_decode_kernel",
           "preprocess_kernel", "fuse_sum_kernel", "heat_argmax_kernel", "conv_f32_kernel", "reproject_kernel", "decode_mem_kernel"]
PACKED = re.compile(r"\bv_pk_(mul|add|fma)_f32\b")


def _code_objects(tmp_path):
    fat = tmp_path / "fatbin.bin"
    subprocess.check_call([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", LIB, str(tmp_path / "discard.so")])
    d = fat.read_bytes()
    magic = b"__CLANG_OFFLOAD_BUNDLE__"
    out, pos = [], 0
    while True:
        i = d.find(magic, pos)
        if i < 0:
            break
        n = struct.unpack_from("<Q", d, i + 24)[0]
        off = i + 32
        for _ in range(n):
            o, sz, tl = struct.unpack_from("<QQQ", d, off)
            off += 24
            triple = d[off:off + tl].decode()
            off += tl
            if "gfx950" in triple and sz:
                p = tmp_path / f"co{len(out)}.elf"
                p.write_bytes(d[i + o:i + o + sz])
                out.append(p)
        pos = i + 24
    return out


def _scan(tmp_path, want_of):
    """(kernel names seen, [(kernel, instruction)] packed-fp32 offenders) over the functions want_of(all functions of a code object) selects."""
    cos = _code_objects(tmp_path)
    assert len(cos) >= 10, "expected one gfx950 code object per translation unit"
    seen, offenders = set(), []
    for co in cos:
        syms = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "--wide", str(co)], capture_output=True, text=True, check=True).stdout
        funcs = [ln.split()[-1] for ln in syms.splitlines() if " FUNC " in ln]
        want = [f for f in want_of(funcs) if not f.endswith(".kd")]
        if not want:
            continue
        dis = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--mcpu=gfx950", f"--disassemble-symbols={','.join(want)}", str(co)],
                             capture_output=True, text=True, check=True).stdout
        cur = None
        for ln in dis.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", ln)
            if m:
                cur = m.group(1)
                seen.add(cur)
                continue
            if cur and PACKED.search(ln):
                offenders.append((cur, ln.strip()))
    return seen, offenders


needs_lib = pytest.mark.skipif(not (os.path.exists(LIB) and os.path.exists(f"{LLVM}/llvm-objdump")), reason="library or llvm tools not present")


@needs_lib
def test_no_packed_fp32_arithmetic_in_bit_exact_kernels(tmp_path):
    seen, offenders = _scan(tmp_path, lambda funcs: [f for f in funcs if any(g in f for g in GUARDED)])
    missing = [g for g in GUARDED if not any(g in s for s in seen)]
    assert not missing, f"guarded kernels not found in the library: {missing}"
    assert not offenders, f"packed fp32 arithmetic in bit-exact kernels: {offenders[:5]} ({len(offenders)} instructions)"


# the f16-MFMA-dense kernels themselves (round 4): the split family's records are compared integer for integer with the fp32 oracle, and the K-split
# A-direct instances exchange fp32 partial accumulators between waves — a f32x4 addition there compiled to v_pk_add_f32
MFMA_FAMILY = ["conv_split_ad_kernel", "conv_f16_kernel", "conv_f16_ad_kernel", "conv_f16_ws_kernel"]


@needs_lib
def test_no_packed_fp32_arithmetic_anywhere_in_the_library(tmp_path):
    """Every device function of every gfx950 code object: the library as shipped holds no v_pk_{mul,add,fma}_f32 at all."""
    seen, offenders = _scan(tmp_path, lambda funcs: funcs)
    missing = [g for g in MFMA_FAMILY + GUARDED if not any(g in s for s in seen)]
    assert not missing, f"kernels not found in the library: {missing}"
    assert len(seen) > 300, f"only {len(seen)} device functions disassembled"
    by_kernel = sorted({k for k, _ in offenders})
    assert not offenders, f"packed fp32 arithmetic in {len(by_kernel)} kernels, e.g. {by_kernel[:3]}: {offenders[:3]} ({len(offenders)} instructions)"

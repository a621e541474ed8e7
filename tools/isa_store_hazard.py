#!/usr/bin/env python3
"""Lint the gfx950 ISA of every kernel source for the store-data hazard described in eagle_amd/csrc/bneck.hip (phase 3):
a buffer/global store of more than 64 bits whose data VGPRs are rewritten by a VALU instruction within the next WAIT issue slots.
hipcc's hazard recognizer covers the case with an immediate soffset only; this script covers the rest.

    python tools/isa_store_hazard.py            # disassembles the gfx950 code objects inside eagle_amd/libeagle_hip.so (what ships; seconds)
    python tools/isa_store_hazard.py --sources  # compiles eagle_amd/csrc/*.hip to assembly under /tmp and scans that instead (minutes)
Exit status 1 on a finding.  tests/test_isa_lint.py runs the first form in the CPU suite.
"""
import glob, os, re, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FLAGS = "--offload-arch=gfx950 -O3 -fno-slp-vectorize -std=c++17 -fPIC -ffp-contract=off -S --cuda-device-only".split()
WAIT = 2                                          # issue slots behind the store that must not write its data registers
STORE = re.compile(r"(buffer|global|flat|scratch)_store_dwordx[34] .*?v\[(\d+):(\d+)\]")


def data_regs(line):
    m = STORE.search(line)
    if not m:
        return None
    # global_store_dwordx4 v[addr], v[data], ...: the DATA operand is the second register range there, the first for buffer_store
    ranges = re.findall(r"v\[(\d+):(\d+)\]", line)
    if m.group(1) == "buffer":
        lo, hi = ranges[0]
    else:
        lo, hi = ranges[1] if len(ranges) > 1 else ranges[0]
    return int(lo), int(hi)


def scan(path):
    lines = open(path).read().split("\n")
    kernel, stores, found = "?", 0, []
    for i, l in enumerate(lines):
        m = re.match(r"^(?:[0-9a-f]+ <)?(_Z\w+)>?:", l)
        if m:
            kernel = m.group(1)
        regs = data_regs(l)
        if regs is None:
            continue
        stores += 1
        lo, hi = regs
        k, j = 0, i + 1
        while k < WAIT and j < len(lines):
            t = lines[j].split("//")[0].strip()
            j += 1
            if not t or t[0] in ";." or t.endswith(":"):
                continue
            k += 1
            if t.startswith("s_endpgm"):
                break
            if t.startswith("s_nop"):
                n = int(t.split()[1]) + 1
                k += n - 1
                continue
            mm = re.match(r"v_\w+\s+v\[?(\d+)(?::(\d+))?\]?", t)
            if mm and not t.startswith("v_cmp") and not t.startswith("v_readlane") and not t.startswith("v_readfirstlane"):
                a = int(mm.group(1)); b = int(mm.group(2) or a)
                if not (b < lo or a > hi):
                    found.append((kernel, i + 1, l.strip(), t))
    return stores, found


def code_objects(so):
    """the gfx950 code objects of every translation unit: .hip_fatbin is a sequence of clang offload bundles (magic, u64 count, {u64 offset, u64 size, u64 len, triple})"""
    import struct
    objcopy = "/opt/rocm/lib/llvm/bin/llvm-objcopy"
    fat = "/tmp/isa_lint/fatbin"
    subprocess.check_call([objcopy, "--dump-section", ".hip_fatbin=" + fat, so, "/tmp/isa_lint/discard.so"])
    blob = open(fat, "rb").read()
    magic, out, pos = b"__CLANG_OFFLOAD_BUNDLE__", [], 0
    while True:
        pos = blob.find(magic, pos)
        if pos < 0:
            break
        (n,) = struct.unpack_from("<Q", blob, pos + 24)
        q = pos + 32
        for _ in range(n):
            off, size, tl = struct.unpack_from("<QQQ", blob, q)
            triple = blob[q + 24:q + 24 + tl].decode()
            q += 24 + tl
            if "gfx950" in triple and size:
                out.append(blob[pos + off:pos + off + size])
        pos += 24
    return out


def main():
    out = "/tmp/isa_lint"
    os.makedirs(out, exist_ok=True)
    total, bad = 0, 0
    if "--sources" not in sys.argv:
        so = os.path.join(ROOT, "eagle_amd", "libeagle_hip.so")
        objs = code_objects(so)
        for k, co in enumerate(objs):
            path = os.path.join(out, f"co{k}.o")
            open(path, "wb").write(co)
            asm = path[:-2] + ".s"
            with open(asm, "w") as f:
                subprocess.check_call(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", "--no-show-raw-insn", path], stdout=f)
            n, found = scan(asm)
            total += n
            for kernel, line, st, nxt in found:
                bad += 1
                print(f"code object {k}:{kernel[:60]} disassembly line {line}: {st}\n    followed by: {nxt}")
        print(f"{total} wide stores scanned in {len(objs)} code objects of {os.path.basename(so)}, {bad} findings")
        return 1 if bad or not total else 0
    srcs = sorted(glob.glob(os.path.join(ROOT, "eagle_amd", "csrc", "*.hip")))
    procs = []
    for s in srcs:
        asm = os.path.join(out, os.path.basename(s)[:-4] + ".s")
        procs.append((s, asm, subprocess.Popen(["/opt/rocm/bin/hipcc", *FLAGS, "-o", asm, s], cwd=os.path.dirname(s), stderr=subprocess.DEVNULL)))
        if len(procs) % 6 == 0:
            for _, _, p in procs[-6:]:
                p.wait()
    for s, asm, p in procs:
        if p.wait() != 0:
            print("compile failed:", s); bad += 1; continue
        n, found = scan(asm)
        total += n
        for kernel, line, st, nxt in found:
            bad += 1
            print(f"{os.path.basename(s)}:{kernel[:60]} asm line {line}: {st}\n    followed by: {nxt}")
    print(f"{total} wide stores scanned in {len(srcs)} sources, {bad} findings")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())

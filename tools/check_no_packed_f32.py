#!/usr/bin/env python3
"""Disassemble the gfx950 code objects inside libmspi_hip.so and list packed-fp32 VALU instructions (v_pk_*_f32).

Why: `v_pk_mul_f32 ... op_sel:[0,1] op_sel_hi:[0,1]` (both result halves read the HIGH half of a source -- what hipcc's
SLP vectoriser emits for broadcast coefficients) returns wrong LOW halves at a rate of ~1e-7 per instruction on MI355X
while another stream's MFMA kernel shares the CU (tools/pk_overlap_probe.hip: 0 errors alone, 8.7k-25k of 1e11 beside an
MFMA kernel, none for the per-half form).  The library is built with -fno-slp-vectorize; this check makes sure no packed
fp32 instruction with a cross-half op_sel slips back in (any v_pk_*_f32 is reported; cross-half ones fail the check).

usage: check_no_packed_f32.py [path/to/libmspi_hip.so]     exit 0 = clean
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(blob):
    """Yield (triple, bytes) for every entry of every clang offload bundle embedded in `blob`."""
    pos = 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return
        base = pos
        n, = struct.unpack_from("<Q", blob, base + len(MAGIC))
        off = base + len(MAGIC) + 8
        for _ in range(n):
            o, size, tl = struct.unpack_from("<QQQ", blob, off)
            triple = blob[off + 24:off + 24 + tl].decode()
            off += 24 + tl
            yield triple, blob[base + o:base + o + size]
        pos = base + len(MAGIC)


def packed_f32(lib):
    blob = open(lib, "rb").read()
    found = []
    n_obj = 0
    for triple, obj in code_objects(blob):
        if "gfx950" not in triple or not obj.startswith(b"\x7fELF"):
            continue
        n_obj += 1
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(obj)
            f.flush()
            asm = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
        func = "?"
        for ln in asm.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:", ln)
            if m:
                func = m.group(1)
            elif re.search(r"\bv_pk_(mul|add|fma)_f32\b", ln):
                found.append((func, ln.strip()))
    return n_obj, found


def cross_half(insn):
    """True when a result half reads the other half of a source: op_sel has a 1 (low result <- high half) or
    op_sel_hi has a 0 on a VGPR-pair source (high result <- low half; harmless for constants / scalar broadcasts,
    which is how the compiler encodes `x * 0.5` -- only the op_sel form is the one measured to fail)."""
    m = re.search(r"op_sel:\[([01,]+)\]", insn)
    return bool(m and "1" in m.group(1))


if __name__ == "__main__":
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                                                             "mspi_amd", "csrc", "libmspi_hip.so")
    n_obj, found = packed_f32(lib)
    bad = [f for f in found if cross_half(f[1])]
    print("%s: %d gfx950 code object(s), %d packed-fp32 instruction(s), %d with a cross-half op_sel" % (lib, n_obj, len(found), len(bad)))
    for func, ln in (bad or found)[:20]:
        print("  %s: %s" % (func, ln))
    sys.exit(1 if bad or n_obj == 0 else 0)

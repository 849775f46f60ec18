"""Build-time disassembly check of the two gfx950 hazards the row kernels work around (csrc/row_common.h: mfa_new,
csrc/vrn_row.hip: rsrc_at) — the workarounds are register-allocation / addressing conventions, so a compiler bump or a
kernel edit could silently undo them; this makes the object code itself the thing that is checked.

    python tools/check_isa.py            # every pcgcv1_amd/lib/obj/*.hip.o, exit 1 on a violation

Rule 1 (MFMA operand overlap).  `v_mfma_f32_4x4x1_16b_f32 vD[4], vA, vB, C`: the 16 blocks of the instruction are
    processed in passes; when D is allocated on top of A or B (possible when the accumulator is NOT tied to C: a fresh
    accumulator whose C is the bias) the later passes read an operand the first pass already overwrote — measured on
    MI355X as wrong values in lanes 12..15 of each 16 when a second wave shares the SIMD.  LLVM marks no early-clobber on
    the 4x4 shapes.  Rule: neither A nor B may lie inside D's register range.  (mfa_new's empty asm keeps a, b and d alive
    together, which forces exactly that.)
Rule 2 (128-bit store data hazard).  A `buffer_store_dwordx4` whose data registers are overwritten by the next VALU
    instruction needs a wait state.  LLVM's hazard recognizer inserts it only when the store has NO register soffset
    (GCNHazardRecognizer::createsVALUHazard); on gfx950 the store loses the race with a register soffset too (measured:
    lanes 12..15 of each 16, one channel).  Rule: no buffer_store_dwordx4 with an SGPR / m0 soffset — the row offset
    travels in the descriptor base instead (rsrc_at).
The compiler version the objects were built with is printed with the result; tests/test_host_cpu.py runs this check.
"""
import glob
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OBJ = os.path.join(ROOT, "pcgcv1_amd", "lib", "obj")
LLVM = os.environ.get("PCGC_LLVM_BIN", "/opt/rocm/lib/llvm/bin")

_MFMA = re.compile(r"^\s*(v_mfma_f32_4x4x1\w*)\s+v\[(\d+):(\d+)\],\s*(\S+?),\s*(\S+?),\s*(\S+)")
_STORE = re.compile(r"^\s*buffer_store_dwordx4\s+v\[\d+:\d+\],\s*(\S+?),\s*s\[\d+:\d+\],\s*(\S+)")
_VREG = re.compile(r"^v(\d+)$")


def disassemble(obj):
    """-> disassembly text of the gfx950 code object bundled in a hipcc host object"""
    d = tempfile.mkdtemp(prefix="pcgc_isa_")
    try:
        src = os.path.join(d, os.path.basename(obj))
        shutil.copy(obj, src)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", src], check=True, capture_output=True)
        dev = [f for f in glob.glob(src + ".*") if "amdgcn" in f]
        if not dev:
            return ""
        return subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", dev[0]], check=True, capture_output=True, text=True).stdout
    finally:
        shutil.rmtree(d, ignore_errors=True)


def check_text(text):
    """-> (number of 4x4x1 MFMAs, number of 128-bit buffer stores, list of violations)"""
    n_mfma = n_store = 0
    bad, func = [], "?"
    for line in text.splitlines():
        if line.endswith(">:") and "<" in line:
            func = line[line.index("<") + 1:-2]
            continue
        m = _MFMA.match(line)
        if m:
            n_mfma += 1
            lo, hi = int(m.group(2)), int(m.group(3))
            for name, op in (("A", m.group(4)), ("B", m.group(5))):
                r = _VREG.match(op)
                if r and lo <= int(r.group(1)) <= hi:
                    bad.append("%s: MFMA operand %s = %s inside D = v[%d:%d]: %s" % (func, name, op, lo, hi, line.split("//")[0].strip()))
            continue
        m = _STORE.match(line)
        if m:
            n_store += 1
            soff = m.group(2)
            if soff.startswith("s") or soff.startswith("m0") or soff.startswith("ttmp"):
                bad.append("%s: 128-bit buffer store with a register soffset: %s" % (func, line.split("//")[0].strip()))
    return n_mfma, n_store, bad


def compiler_version():
    try:
        out = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--version"], capture_output=True, text=True).stdout
        return " / ".join(l.strip() for l in out.splitlines() if "version" in l.lower())[:200]
    except OSError:
        return "unknown"


def main():
    objs = sorted(glob.glob(os.path.join(OBJ, "*.hip.o")))
    if not objs:
        print("no objects under %s: run `python -m pcgcv1_amd.build` first" % OBJ)
        return 2
    total_bad = []
    for obj in objs:
        n_mfma, n_store, bad = check_text(disassemble(obj))
        print("%-22s %6d 4x4x1 MFMAs, %5d 128-bit buffer stores, %d violations" % (os.path.basename(obj), n_mfma, n_store, len(bad)))
        total_bad += ["%s: %s" % (os.path.basename(obj), b) for b in bad]
    print("compiler:", compiler_version())
    for b in total_bad[:40]:
        print("VIOLATION", b)
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())

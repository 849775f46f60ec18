"""Build-time disassembly check of the gfx950 store-data hazard the row kernels work around (csrc/vrn_row.hip: rsrc_at;
csrc/row_common.h: mfa_new) — the workaround is an addressing convention, so a compiler bump or a kernel edit could
silently undo it; this makes the object code itself the thing that is checked.

    python tools/check_isa.py            # every pcgcv1_amd/lib/obj/*.hip.o, exit 1 on a violation

The hazard (round 2: one channel of lanes 12..15 of each row of 16 wrong, run to run different, only when two waves share
a SIMD).  A `buffer_store_dwordx4` reads its four data registers over several cycles; an instruction right behind it that
WRITES one of them (a VALU op — or the first MFMA of a fresh accumulator the register allocator placed on the registers
just stored) can win that race.  LLVM's hazard recognizer inserts the wait state only when the store has NO register
soffset (GCNHazardRecognizer::createsVALUHazard: "this hazard only exists if the instruction is not using a register in
the soffset field" — true for the parts that comment was written for); gfx950 loses the race with a register soffset too.
Both round-2 symptoms have this one cause: `mfa_new` (keeps a fresh accumulator off the registers of its operands) moved
the allocation so that the new accumulator no longer landed on just-stored data, `rsrc_at` (row offset in the descriptor
base, soffset = 0) removed the cause by making every 128-bit store one the compiler protects.
  ERROR  rule: no `buffer_store_dwordx4` with an SGPR / m0 soffset anywhere in the library.
  INFO   `v_mfma_f32_4x4x1_16b_f32` whose destination overlaps its A or B operand.  Suspected in round 2, cleared in round 3:
         tools/exp/exp_mfma_overlap.hip runs the instruction with D on top of A, B or both (every position, abid 0 / 5 / 15,
         C a register quad or the literal 0, isolated / followed / surrounded by independent MFMAs, 1 / 2 / 4 waves per
         SIMD): 0 mismatches in 324 variants x 5e8 - 2e9 lane-iterations on MI355X (profiles/r03_mfma_overlap.txt).  The
         shipped objects contain such overlaps (the count is printed) in kernels that pass the bit-exact slot-invariance
         and 200-step soak tests; they are reported, not refused.
The compiler version the objects were built with is printed with the result; tests/test_host_cpu.py runs this check, and
tests/test_gpu_parity.py::test_every_row_kernel_is_slot_invariant_and_repeatable is the run-time side of it.
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
    """-> (number of 4x4x1 MFMAs, number of 128-bit buffer stores, list of violations, list of D / operand overlaps)"""
    n_mfma = n_store = 0
    bad, overlaps, func = [], [], "?"
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
                    overlaps.append("%s: MFMA operand %s = %s inside D = v[%d:%d]: %s" % (func, name, op, lo, hi, line.split("//")[0].strip()))
            continue
        m = _STORE.match(line)
        if m:
            n_store += 1
            soff = m.group(2)
            if soff.startswith("s") or soff.startswith("m0") or soff.startswith("ttmp"):
                bad.append("%s: 128-bit buffer store with a register soffset: %s" % (func, line.split("//")[0].strip()))
    return n_mfma, n_store, bad, overlaps


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
        n_mfma, n_store, bad, overlaps = check_text(disassemble(obj))
        print("%-22s %6d 4x4x1 MFMAs (%d with D over A / B: informational), %5d 128-bit buffer stores, %d violations"
              % (os.path.basename(obj), n_mfma, len(overlaps), n_store, len(bad)))
        total_bad += ["%s: %s" % (os.path.basename(obj), b) for b in bad]
    print("compiler:", compiler_version())
    for b in total_bad[:40]:
        print("VIOLATION", b)
    return 1 if total_bad else 0


if __name__ == "__main__":
    sys.exit(main())

"""Instruction mix per kernel of a built object:  python tools/isa_mix.py vrn_row [substring ...]
(MFMA / VALU / SALU / s_waitcnt / branches / loads / stores / LDS / s_nop per kernel symbol; code bytes)"""
import collections
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import check_isa  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("s_waitcnt"):
        return "wait"
    if op.startswith("s_nop"):
        return "nop"
    if op.startswith("s_cbranch") or op.startswith("s_branch"):
        return "br"
    if op.startswith("s_barrier"):
        return "bar"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("buffer_load") or op.startswith("global_load"):
        return "vld"
    if op.startswith("buffer_store") or op.startswith("global_store"):
        return "vst"
    if op.startswith("ds_"):
        return "lds"
    return "other"


def main():
    name = sys.argv[1]
    subs = sys.argv[2:]
    obj = os.path.join(ROOT, "pcgcv1_amd", "lib", "obj", name + ".hip.o") if not name.endswith(".o") else name
    txt = check_isa.disassemble(obj)
    for s in re.split(r"\n(?=[0-9a-f]{16} <)", txt):
        m = re.match(r"([0-9a-f]{16}) <([^>]+)>", s)
        if not m:
            continue
        sym = m.group(2)
        if subs and not all(x in sym for x in subs):
            continue
        lines = [ln for ln in s.split("\n")[1:] if ln.strip()]
        ops = collections.Counter(classify(ln.split()[0]) for ln in lines if ln.split())
        addrs = [int(mm.group(1), 16) for mm in (re.search(r"//\s*([0-9A-Fa-f]{12}):", ln) for ln in lines) if mm]
        size = (max(addrs) - min(addrs)) if addrs else 0
        try:
            import subprocess
            dem = subprocess.run(["c++filt", sym], capture_output=True, text=True).stdout.strip()
        except Exception:
            dem = sym
        print("%-100s %6d B  %s" % (dem[:100], size, " ".join("%s=%d" % kv for kv in sorted(ops.items()))))


if __name__ == "__main__":
    main()

"""Condensed GPU timeline of the tail of a rocprofv3 (rocpd) kernel trace: consecutive launches of the same kernel
family on the same queue are merged into one line (start, end, busy time, count), idle gaps > min_us are printed
inline.    python tools/rocpd_timeline.py trace_results.db [tail_ms] [min_gap_us]"""
import re
import sqlite3
import sys


def family(n):
    n = n.split("(")[0]
    n = re.sub(r"<.*", "", n).replace("void ", "").replace("pcgc::", "")
    if n.startswith("at::native") or "elementwise" in n:
        return "torch_elementwise"
    return n[-40:]


def main(path, tail_ms=130.0, min_gap_us=100.0):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    q = "queue_id" if "queue_id" in cols else ("stream_id" if "stream_id" in cols else "0")
    rows = [(s, e, family(n), qq) for s, e, n, qq in c.execute("select start, end, name, %s from kernels" % q)]
    rows.sort()
    t1 = rows[-1][1]
    rows = [r for r in rows if r[0] > t1 - tail_ms * 1e6]
    t0 = rows[0][0]
    runs = []
    for s, e, f, qq in rows:
        if runs and runs[-1][2] == f and runs[-1][3] == qq:
            runs[-1][1] = max(runs[-1][1], e); runs[-1][4] += e - s; runs[-1][5] += 1
        else:
            runs.append([s, e, f, qq, e - s, 1])
    cur_end = t0
    for s, e, f, qq, busy, n in runs:
        if s - cur_end > min_gap_us * 1e3:
            print("            ---- idle %.2f ms ----" % ((s - cur_end) / 1e6))
        print("%8.2f %8.2f  q%-3s %-40s x%-4d busy %.2f ms" % ((s - t0) / 1e6, (e - t0) / 1e6, qq, f, n, busy / 1e6))
        cur_end = max(cur_end, e)


if __name__ == "__main__":
    main(sys.argv[1], *(float(a) for a in sys.argv[2:]))

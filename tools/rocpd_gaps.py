"""GPU idle gaps in a rocprofv3 (rocpd) kernel trace: for the last `--steps` benchmark steps print every gap between
consecutive kernel/copy executions that exceeds --min-us, with the kernels on either side, plus busy/idle totals.
    python tools/rocpd_gaps.py trace_results.db [min_us]"""
import sqlite3
import sys


def main(path, min_us=100.0):
    c = sqlite3.connect(path)
    tables = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    rows = []
    if "kernels" in tables:
        rows += [(s, e, n) for s, e, n in c.execute("select start, end, name from kernels")]
    if "memory_copies" in tables:
        try:
            rows += [(s, e, "memcpy:" + str(n)) for s, e, n in c.execute("select start, end, name from memory_copies")]
        except sqlite3.OperationalError:
            pass
    rows.sort()
    if not rows:
        print("no kernels; tables:", tables)
        return
    t0, t1 = rows[0][0], rows[-1][1]
    busy, cur_end = 0, rows[0][0]
    gaps = []
    for s, e, n in rows:
        if s > cur_end:
            gaps.append((s - cur_end, cur_end, prev, n))
            busy += e - s
        else:
            busy += max(0, e - cur_end)
        if e > cur_end:
            cur_end, prev = e, n
    print("span %.1f ms, busy %.1f ms, idle %.1f ms, %d launches" % ((t1 - t0) / 1e6, busy / 1e6, (t1 - t0 - busy) / 1e6, len(rows)))
    # last second of the trace = steady state
    tail = [g for g in gaps if g[1] > t1 - 0.6e9]
    print("gaps > %.0f us in the last 600 ms:" % min_us)
    for d, at, a, b in tail:
        if d / 1e3 >= min_us:
            print("  %8.1f us  at %9.2f ms  after %-60s before %s" % (d / 1e3, (at - t0) / 1e6, a.split("(")[0][-60:], b.split("(")[0][-60:]))
    print("idle in last 600 ms: %.1f ms in %d gaps" % (sum(g[0] for g in tail) / 1e6, len(tail)))
    # where the idle time sits: by gap size and by the launch that follows the gap
    edges = [0, 20e3, 50e3, 100e3, 300e3, 1e6, 1e12]
    for lo, hi in zip(edges[:-1], edges[1:]):
        sel = [g[0] for g in tail if lo <= g[0] < hi]
        print("  gaps %6.0f..%-8.0f us: %4d, %6.2f ms" % (lo / 1e3, hi / 1e3, len(sel), sum(sel) / 1e6))
    by = {}
    for d, at, a, b in tail:
        k = b.split("(")[0][-50:]
        by[k] = by.get(k, 0) + d
    for k, v in sorted(by.items(), key=lambda kv: -kv[1])[:12]:
        print("  idle before %-52s %6.2f ms" % (k, v / 1e6))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 100.0)

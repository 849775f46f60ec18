"""Per-kernel mean of every PMC counter in one or more rocprofv3 (rocpd) result databases.
    python tools/rocpd_pmc.py a.db b.db ... > profiles/rNN_pmc.csv
Values are per dispatch (mean over dispatches of that kernel).  FETCH_SIZE / WRITE_SIZE are in KiB as
rocprofv3 reports them; on gfx950 FETCH_SIZE counts 128-B requests as 64 B for wide coalesced reads
(MI355X_MICROARCH.md §HBM), so FETCH_x2_MB doubles it — the upper-bound estimate of bytes fetched."""
import sqlite3
import sys
from collections import defaultdict


def main(paths):
    agg = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    dur = defaultdict(lambda: [0.0, 0])
    for p in paths:
        c = sqlite3.connect(p)
        for name, counter, value, d in c.execute("select kernel_name, counter_name, value, duration from counters_collection"):
            a = agg[name][counter]
            a[0] += value
            a[1] += 1
            dur[name][0] += d
            dur[name][1] += 1
    counters = sorted({k for v in agg.values() for k in v})
    print("Kernel,Dispatches,AvgDurationNs," + ",".join(counters) + ",FETCH_x2_MB,WRITE_MB")
    rows = sorted(agg.items(), key=lambda kv: -dur[kv[0]][0])
    for name, cs in rows:
        n = max(v[1] for v in cs.values())
        vals = [cs[k][0] / cs[k][1] if k in cs else float("nan") for k in counters]
        f = cs["FETCH_SIZE"][0] / cs["FETCH_SIZE"][1] * 2 * 1024 / 1e6 if "FETCH_SIZE" in cs else float("nan")
        w = cs["WRITE_SIZE"][0] / cs["WRITE_SIZE"][1] * 1024 / 1e6 if "WRITE_SIZE" in cs else float("nan")
        print('"%s",%d,%.0f,%s,%.2f,%.2f' % (name.split("(")[0][:90], n, dur[name][0] / dur[name][1],
                                             ",".join("%.1f" % v for v in vals), f, w))


if __name__ == "__main__":
    main(sys.argv[1:])

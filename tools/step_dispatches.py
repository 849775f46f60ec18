"""Dispatches of ONE training step from a rocprofv3 --kernel-trace csv of tools/bench_train.py: the kernels between two
consecutive adam_kernel launches, by name.
    python tools/step_dispatches.py <dir with *_kernel_trace.csv> [*_memory_copy_trace.csv is counted when present]
(A run's kernel_stats divide ALL dispatches by the steps and so count the set-up's uploads too.)"""
import collections
import csv
import glob
import os
import sys


def main(d):
    kt = sorted(glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True))
    if not kt:
        sys.exit("no *kernel_trace.csv under " + d)
    rows = list(csv.DictReader(open(kt[0])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    marks = [i for i, r in enumerate(rows) if "adam_kernel" in r["Kernel_Name"]]
    copies = []
    for f in glob.glob(os.path.join(d, "**", "*memory_copy_trace.csv"), recursive=True):
        copies += [int(r["Start_Timestamp"]) for r in csv.DictReader(open(f))]
    last = None
    for n, (a, b) in enumerate(zip(marks, marks[1:]), 1):
        step = rows[a + 1:b + 1]
        t0, t1 = int(rows[a]["End_Timestamp"]), int(rows[b]["End_Timestamp"])
        kinds = collections.Counter("at::native (torch)" if "at::native" in r["Kernel_Name"] else "pcgc kernels" for r in step)
        nc = sum(1 for t in copies if t0 < t <= t1)
        if nc:
            kinds["copies"] = nc
        print("step %d: %d dispatches in %.2f ms: %s" % (n, len(step) + nc, (t1 - t0) / 1e6, dict(kinds)))
        last = step
    if last:
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last)
        # time no kernel is running (kernels of one stream: overlaps are launch tails of back-to-back dispatches)
        idle, end = 0, int(last[0]["Start_Timestamp"])
        for r in last:
            idle += max(0, int(r["Start_Timestamp"]) - end)
            end = max(end, int(r["End_Timestamp"]))
        gaps, end, prev = [], int(last[0]["Start_Timestamp"]), "(step start)"
        for r in last:
            g = int(r["Start_Timestamp"]) - end
            if g > 0:
                gaps.append((g, prev, r["Kernel_Name"].split("(")[0]))
            if int(r["End_Timestamp"]) > end:
                end, prev = int(r["End_Timestamp"]), r["Kernel_Name"].split("(")[0]
        print("# largest gaps of the last step (us, after -> before):")
        for g, a_, b_ in sorted(gaps, reverse=True)[:12]:
            print("%8.1f  %s -> %s" % (g / 1e3, a_[-60:], b_[-60:]))
        print("# last step: kernel durations sum to %.2f ms, %.2f ms with no kernel running; by kernel (launches, us in the step):" % (busy / 1e6, idle / 1e6))
        c, t = collections.Counter(), collections.Counter()
        for r in last:
            k = r["Kernel_Name"].split("(")[0]
            c[k] += 1
            t[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        for k, v in t.most_common():
            print("%3d %8.1f  %s" % (c[k], v / 1e3, k))


if __name__ == "__main__":
    main(sys.argv[1])

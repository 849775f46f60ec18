"""Summarise a rocprofv3 (ROCm 7.2, rocpd SQLite output) kernel trace into the per-kernel stats table that
`--stats` reports: name, calls, total ns, average ns, percentage, plus LDS / VGPR per kernel.

    python tools/rocpd_stats.py gpurun_out/prof/x_results.db > profiles/rNN_name_kernel_stats.csv
"""
import sqlite3
import sys


def main(path):
    c = sqlite3.connect(path)
    rows = c.execute(
        "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), max(lds_size), "
        "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(grid_x), max(workgroup_x) from kernels "
        "group by name order by sum(duration) desc").fetchall()
    total = sum(r[2] for r in rows) or 1
    print("Name,Calls,TotalDurationNs,AverageNs,MinNs,MaxNs,Percentage,LDS,VGPR,AGPR,SGPR,MaxGridX,WorkgroupX")
    for r in rows:
        print('"%s",%d,%d,%.0f,%d,%d,%.2f,%d,%d,%d,%d,%d,%d' % (r[0], r[1], r[2], r[3], r[4], r[5], 100.0 * r[2] / total,
                                                          r[6] or 0, r[7] or 0, r[8] or 0, r[9] or 0, r[10] or 0, r[11] or 0))


if __name__ == "__main__":
    main(sys.argv[1])

#!/usr/bin/env python
"""Summarises a rocprofv3 (rocpd sqlite) kernel trace into a small CSV for profiles/.

usage: tools/rocprof_summary.py gpurun_out/prof_rNN/xxx_results.db profiles/rNN_kernel_stats.csv
"""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    cur = db.cursor()
    rows = list(cur.execute(
        "select name, count(*), sum(end-start), avg(end-start), min(end-start), max(end-start), "
        "max(vgpr_count), max(sgpr_count), max(lds_size), max(scratch_size), max(grid_x), max(workgroup_x) "
        "from kernels group by name order by sum(end-start) desc"))
    total = sum(r[2] for r in rows) or 1
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "MinNs", "MaxNs", "Percentage", "VGPR", "SGPR",
                    "LDS_bytes", "Scratch_bytes", "max_grid_x", "workgroup_x"])
        for r in rows:
            w.writerow([r[0], r[1], int(r[2]), "%.1f" % r[3], int(r[4]), int(r[5]), "%.3f" % (100.0 * r[2] / total)]
                       + list(r[6:]))
    print(open(out_path).read())


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

#!/usr/bin/env python
"""Turn a rocprofv3 results database (rocprofv3 --kernel-trace --stats ... writes <name>_results.db with ROCm 7.2)
into the per-kernel summary committed under profiles/.  usage: rocprof_summary.py results.db out.csv"""
import csv
import sqlite3
import sys


def main(db_path, out_path):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    extra = {}
    for name, vg, ag, sg, lds, gx, wx in db.execute(
            "select name, max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) from kernels group by name"):
        extra[name] = (vg, ag, sg, lds, gx, wx)
    with open(out_path, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage", "VGPR", "AGPR", "SGPR", "LDS_bytes", "max_grid_x", "workgroup_x"])
        for name, calls, tot, avg, pct in rows:
            e = extra.get(name, ("",) * 6)
            w.writerow([name, calls, round(tot, 3), round(avg, 3), round(pct, 3)] + list(e))
    print("wrote", out_path, len(rows), "kernels")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

#!/usr/bin/env python
"""Turn a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME` writes
NAME_results.db on ROCm 7.2) into the per-kernel stats CSV committed under profiles/."""
import csv
import sqlite3
import sys


def main(db_path, out_csv):
    db = sqlite3.connect(db_path)
    rows = list(db.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    with open(out_csv, "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationUs", "AverageUs", "Percentage"])
        for r in rows:
            w.writerow([r[0], r[1], "%.3f" % r[2], "%.3f" % r[3], "%.3f" % r[4]])
    print("wrote %s (%d kernels)" % (out_csv, len(rows)))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])

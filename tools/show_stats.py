#!/usr/bin/env python
"""Print a kernel-stats CSV (tools/rocpd_stats.py) as a table: calls, average us, share of the kernel time."""
import csv
import sys

rows = list(csv.DictReader(l for l in open(sys.argv[1]) if not l.startswith("#")))
tot = sum(float(r['TotalDurationUs']) for r in rows)
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
for r in rows[:n]:
    name = r['Name'].replace('stattn::(anonymous namespace)::', '').replace('void ', '')
    print('%-78s %6s x %9.1f us %5.1f%%' % (name[:78], r['Calls'], float(r['AverageUs']), 100 * float(r['TotalDurationUs']) / tot))
print('total kernel time %.2f ms' % (tot / 1e3))

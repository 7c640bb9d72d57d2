#!/usr/bin/env python
"""Print the kernel sequence of a rocprofv3 rocpd database between two dispatch indices: start offset (us) from the first
printed kernel, duration, gap to the previous kernel's end.  usage: seq_dump.py <db> <first> <count>"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table', 'view')")]
src = next((n for n in names if n == "kernels"), None) or next((n for n in names if "kernel_dispatch" in n), None)
cols = [r[1] for r in db.execute("pragma table_info(%s)" % src)]
s_col = next(c for c in cols if c.lower() in ("start", "start_timestamp"))
e_col = next(c for c in cols if c.lower() in ("end", "end_timestamp"))
n_col = next(c for c in cols if c.lower() in ("name", "kernel_name"))
rows = sorted(db.execute("select %s, %s, %s from %s" % (s_col, e_col, n_col, src)))
first, count = int(sys.argv[2]), int(sys.argv[3])
if first < 0:
    first = len(rows) + first
t0, prev = rows[first][0], rows[first][0]
for s, e, nm in rows[first:first + count]:
    nm = str(nm).replace("stattn::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:46]
    print("%9.1f us  +%6.1f gap  %7.1f us  %s" % ((s - t0) / 1e3, (s - prev) / 1e3, (e - s) / 1e3, nm))
    prev = e

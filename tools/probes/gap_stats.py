#!/usr/bin/env python
"""Idle time between kernels in a rocprofv3 rocpd database: how much of the wall span of the kernel stream is not covered
by any kernel.  usage: gap_stats.py <db> [skip_fraction]  (the first `skip_fraction` of the dispatches -- warm-up, set-up --
is ignored; default 0.5)"""
import sqlite3
import sys


def main(db_path, skip=0.5):
    db = sqlite3.connect(db_path)
    names = [r[0] for r in db.execute("select name from sqlite_master where type in ('table', 'view')")]
    src = next((n for n in names if n == "kernels"), None) or next((n for n in names if "kernel_dispatch" in n), None)
    if src is None:
        print("no kernel table among", names)
        return
    cols = [r[1] for r in db.execute("pragma table_info(%s)" % src)]
    s_col = next(c for c in cols if c.lower() in ("start", "start_timestamp"))
    e_col = next(c for c in cols if c.lower() in ("end", "end_timestamp"))
    n_col = next((c for c in cols if c.lower() in ("name", "kernel_name")), None)
    rows = sorted(db.execute("select %s, %s, %s from %s" % (s_col, e_col, n_col or "''", src)))
    rows = rows[int(len(rows) * skip):]
    busy, cover_end, gaps, prev, by_pair = 0, rows[0][0], [], "", {}
    for s, e, nm in rows:
        nm = str(nm).replace("stattn::(anonymous namespace)::", "").replace("void ", "").split("(")[0][:40]
        if s > cover_end:
            gaps.append(s - cover_end)
            k = (prev, nm)
            c = by_pair.setdefault(k, [0, 0])
            c[0] += 1; c[1] += s - cover_end
        busy += max(0, e - max(s, cover_end))
        cover_end = max(cover_end, e)
        prev = nm
    span = cover_end - rows[0][0]
    gaps.sort()
    print("%d dispatches, span %.3f ms, covered by kernels %.3f ms (%.1f %%), %d gaps: median %.2f us, mean %.2f us, p95 %.2f us, largest %.1f us"
          % (len(rows), span / 1e6, busy / 1e6, 100.0 * busy / span, len(gaps), gaps[len(gaps) // 2] / 1e3, sum(gaps) / len(gaps) / 1e3,
             gaps[int(len(gaps) * 0.95)] / 1e3, gaps[-1] / 1e3))
    print("gaps by (kernel before -> kernel after), by total idle time:")
    for (a, b), (cnt, tot) in sorted(by_pair.items(), key=lambda kv: -kv[1][1])[:25]:
        print("  %6d x %8.2f us avg  = %8.2f ms   %s -> %s" % (cnt, tot / cnt / 1e3, tot / 1e6, a, b))


if __name__ == "__main__":
    main(sys.argv[1], float(sys.argv[2]) if len(sys.argv) > 2 else 0.5)

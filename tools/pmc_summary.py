#!/usr/bin/env python
"""Aggregate a rocprofv3 `--pmc ... --output-format csv` counter_collection file per kernel:
mean counter value per launch.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE counts 64 B per
128-B request for wide coalesced reads, so the read bytes are reported doubled as well
(MI355X_MICROARCH.md, HBM section)."""
import collections
import csv
import sys


def main(path, out=None):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(path)):
        agg[r['Kernel_Name']][r['Counter_Name']].append(float(r['Counter_Value']))
    lines = ["kernel,counter,launches,mean_per_launch,MB_per_launch(raw KiB->MB),MB_per_launch(x2 gfx950 read correction)"]
    for k in sorted(agg, key=lambda k: -sum(sum(v) for v in agg[k].values())):
        for c, v in sorted(agg[k].items()):
            mean = sum(v) / len(v)
            mb = mean * 1024 / 1e6 if c in ('FETCH_SIZE', 'WRITE_SIZE') else float('nan')
            lines.append('"%s",%s,%d,%.4g,%.2f,%s' % (k[:110], c, len(v), mean, mb, ("%.2f" % (2 * mb)) if c == 'FETCH_SIZE' else ""))
    txt = "\n".join(lines) + "\n"
    if out:
        open(out, "w").write(txt)
    else:
        sys.stdout.write(txt)


if __name__ == "__main__":
    main(*sys.argv[1:3])

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


def bench_traffic(fetch_csv, write_csv, out_json, keys):
    """profiles/pmc_traffic.json for bench.py: HBM-side bytes per launch (FETCH_SIZE x 2 + WRITE_SIZE) of the kernels its
    roofline objects name, from two summaries written by main().  `keys`: the bench configurations the PMC command covers.
    A class may span several kernel symbols (the plain forward GEMM launches are gemm2_group_kernel and gemm2_kernel
    instances): launch-weighted mean over all of them."""
    import json
    pats = {"gemm_nn": ["gemm2_group_kernel<1, 1, false, false, false", "gemm2_group_kernel<2, 2, false, false, false",
                        "gemm2_kernel<1, 1, false, false, false"],
            "spatial": ["spatial2_kernel<128>"], "temporal": ["temporal_kernel"]}

    def total(path, plist, idx):
        tot, n = 0.0, 0
        for r in csv.reader(open(path)):
            if len(r) > idx and any(p in r[0] for p in plist):
                tot += float(r[idx]) * 1e6 * int(r[2]); n += int(r[2])
        return tot, n
    vals = {}
    for k, plist in pats.items():
        f, n = total(fetch_csv, plist, 5)
        w, _ = total(write_csv, plist, 4)
        vals[k] = (f + w) / n if n else None
    try:
        cur = json.load(open(out_json))
    except Exception:
        cur = {}
    for k in keys:
        cur[k] = vals
    json.dump(cur, open(out_json, "w"), indent=1, sort_keys=True)
    print(vals)

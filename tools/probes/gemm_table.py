#!/usr/bin/env python
"""Per-launch table of the GEMM classes of bench.py JSON lines: usage gemm_table.py a.json [b.json ...] (one column per file)"""
import json, sys
ds = []
for p in sys.argv[1:]:
    txt = [l for l in open(p) if l.startswith('{')]
    ds.append(json.loads(txt[-1]))
keys = [k for k in ds[0]['kernels'] if 'gemm' in k]
print("%-60s" % "launch" + "".join("%22s" % p.split('/')[-1][:20] for p in sys.argv[1:]))
tot = [0.0] * len(ds)
for k in keys:
    row = "%-60s" % k[:58]
    for i, d in enumerate(ds):
        v = d['kernels'].get(k)
        if v:
            row += "%10.1f us %6.1f TF" % (1e3 * v['ms_per_launch'], v.get('achieved', 0)); tot[i] += 1e3 * v['ms_per_launch']
        else: row += "%22s" % "-"
    print(row)
print("%-60s" % "ms_per_step" + "".join("%22.3f" % d['ms_per_step'] for d in ds))

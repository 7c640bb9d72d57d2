#!/usr/bin/env python
"""Memory / wait / barrier skeleton of one kernel with source lines.  usage: isa_skeleton.py listing.s <kernel name substring>
(listing from `hipcc ... -gline-tables-only --cuda-device-only -S`).  Prints every global / LDS / scalar load and store, s_waitcnt,
s_barrier, ds_bpermute, scratch access and branch of the kernel in layout order, runs collapsed, each with the source line it came
from: how the serial `ds_bpermute + s_waitcnt` chains of the wave reductions and the load-wait-load-wait groups of
spatial_bwd_kernel were found."""
import re, sys
src, key = sys.argv[1], sys.argv[2]
lines = open(src).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + re.escape(key) + r'\S*:', l))
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
    if m: files[int(m.group(1))] = m.group(2).split('/')[-1]
    m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]+)"\s+md5', l)
    if m: files[int(m.group(1))] = m.group(2).split('/')[-1]
cur = None; out = []
for l in lines[start:]:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = "%s:%s" % (files.get(int(m.group(1)), m.group(1)), m.group(2)); continue
    t = l.strip()
    if t.startswith('s_endpgm'): break
    if re.match(r'(\.LBB\S+:)', t): out.append(('', t.split()[0])); continue
    if re.match(r'(global_load|global_store|buffer_load|buffer_store|s_waitcnt vmcnt|s_waitcnt lgkmcnt|s_barrier|ds_bpermute|scratch_|s_cbranch|ds_read|ds_write|v_mfma|s_load)', t):
        op = t.split()[0] + (' ' + t.split()[1] if t.startswith(('s_waitcnt', 's_cbranch')) else '')
        out.append((cur, op))
prev = None; cnt = 0
for c, t in out:
    if (c, t) == prev: cnt += 1
    else:
        if prev: print("%-22s %s%s" % (prev[0], prev[1], ' x%d' % cnt if cnt > 1 else ''))
        prev = (c, t); cnt = 1
print("%-22s %s%s" % (prev[0], prev[1], ' x%d' % cnt if cnt > 1 else ''))

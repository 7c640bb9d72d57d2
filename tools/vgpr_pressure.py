#!/usr/bin/env python
"""Where a kernel's VGPR pressure peaks.  usage: vgpr_pressure.py listing.s <kernel name substring> [n]
(listing from `hipcc ... -gline-tables-only --cuda-device-only -S`).  Straight-line approximation: a value is live from the instruction
that writes its register to the last one that reads it before the next write (layout order, control flow ignored -- the attention kernels
are one long block per item);
prints the n instructions with the most registers live and the source lines that hold the most registers alive around the peak."""
import re, sys
src, key = sys.argv[1], sys.argv[2]
n = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lines = open(src).read().splitlines()
start = next(i for i, l in enumerate(lines) if re.match(r'^_Z\S*' + re.escape(key) + r'\S*:', l))
files = {}
for l in lines:
    m = re.match(r'\s*\.file\s+(\d+)\s+"[^"]*"\s+"([^"]+)"', l)
    if m: files[int(m.group(1))] = m.group(2).split('/')[-1]
ins = []; cur = None
for l in lines[start + 1:]:
    m = re.match(r'\s*\.loc\s+(\d+)\s+(\d+)', l)
    if m: cur = "%s:%s" % (files.get(int(m.group(1)), m.group(1)), m.group(2)); continue
    t = l.strip()
    if not t or t.startswith(('.', ';')) or t.endswith(':'): continue
    if t.startswith('s_endpgm'): break
    regs = set()
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', t): regs.update(range(int(a), int(b) + 1))
    for a in re.findall(r'\bv(\d+)\b', t): regs.add(int(a))
    ins.append((cur, t, regs))
# value live ranges: a write (first operand of anything that is not a store / compare / LDS write / export) starts a new value of the
# register, a read extends the current one
def regs_of(tok):
    out = set()
    for a, b in re.findall(r'\bv\[(\d+):(\d+)\]', tok): out.update(range(int(a), int(b) + 1))
    for a in re.findall(r'\bv(\d+)\b', tok): out.add(int(a))
    return out
nodef = ('global_store', 'buffer_store', 'scratch_store', 'ds_write', 'ds_bpermute_b32_nodst', 'v_cmp', 'v_cmpx', 's_', 'flat_store', 'global_atomic')
live = [0] * len(ins)
cur_start, cur_last, cur_line, first, defline = {}, {}, {}, {}, {}
ranges = []
def close(r):
    if r in cur_start: ranges.append((r, cur_start[r], cur_last[r], cur_line[r]))
for i, (c, t, regs) in enumerate(ins):
    mn = t.split()[0]
    ops = t[len(mn):].split(',')
    dst = regs_of(ops[0]) if ops and not mn.startswith(nodef) else set()
    srcs = regs_of(','.join(ops[1:])) if dst else regs
    if mn.startswith(('v_fmac', 'v_mac', 'v_pk_fma', 'v_mfma', 'v_dot')) or 'dpp' in t or 'v_cndmask' in mn: srcs = srcs | dst      # (read-modify-write forms)
    for r in srcs:
        if r in cur_start: cur_last[r] = i
        else: cur_start[r] = i; cur_last[r] = i; cur_line[r] = c          # (read before any write: live-in)
    for r in dst:
        if r in srcs and r in cur_start: cur_last[r] = i; continue
        close(r)
        cur_start[r] = i; cur_last[r] = i; cur_line[r] = c
for r in list(cur_start): close(r)
for r, a, b, c in ranges:
    for i in range(a, b + 1): live[i] += 1
first = {r for r, _, _, _ in ranges}
order = sorted(range(len(ins)), key=lambda i: -live[i])
print("registers named: %d; peak %d live at instruction %d of %d" % (len(first), live[order[0]], order[0], len(ins)))
seen = set()
for i in order:
    if any(abs(i - j) < 40 for j in seen): continue
    seen.add(i)
    print("  %4d live at #%d  %-22s %s" % (live[i], i, ins[i][0], ins[i][1][:70]))
    if len(seen) >= n: break
pk = order[0]
held = {}
for r, a, b, c in ranges:
    if a <= pk <= b: held[c] = held.get(c, 0) + 1
print("registers alive at the peak, by the source line that first named them:")
for c, k in sorted(held.items(), key=lambda x: -x[1])[:14]: print("  %3d  %s" % (k, c))
if len(sys.argv) > 4:      # detail: the value ranges of one defining source line that cross the peak
    want = sys.argv[4]
    for r, a, b, c in sorted(ranges, key=lambda x: x[1]):
        if a <= pk <= b and c and c.endswith(want):
            print("  v%-3d written at #%d (%s), last read at #%d (%s): %s" % (r, a, ins[a][0], b, ins[b][0], ins[b][1][:60]))

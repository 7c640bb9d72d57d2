#!/usr/bin/env python
"""Which kernels changed between two builds?  usage: kernel_code_diff.py old.s new.s   (listings from `hipcc ... --cuda-device-only -S`)
Hashes the instruction stream of every kernel in two listings (labels renumbered) and lists the kernels whose code differs, the new
ones and the removed ones.  Used in round 5 to show that the kernels profiled at dce9137 are instruction-for-instruction the ones HEAD
ships (later commits only ADD kernels that sit behind switches): DESIGN.md section 6."""
import sys, re, hashlib
# per-kernel hash of the instruction text in two `hipcc -S` listings (labels renumbered: the function index in .LBBn_m shifts when functions are added)
def kernels(path):
    res = {}; cur = None; buf = []
    for l in open(path):
        m = re.match(r'^(_Z\S+):\s', l)
        if m:
            cur = m.group(1); buf = []; continue
        if cur is None: continue
        t = l.split(';')[0].strip()
        if t.startswith('s_endpgm'):
            buf.append(t); res[cur] = hashlib.sha1('\n'.join(buf).encode()).hexdigest()[:12]; cur = None; continue
        if not t or t.startswith('.') and not t.startswith('.LBB'): continue
        t = re.sub(r'\.LBB\d+_', '.LBB_', t)
        buf.append(t)
    return res
a, b = kernels(sys.argv[1]), kernels(sys.argv[2])
same = [k for k in a if k in b and a[k] == b[k]]
diff = [k for k in a if k in b and a[k] != b[k]]
print("%s -> %s: kernels %d -> %d; identical instruction stream: %d; changed: %d; new: %d; gone: %d" % (sys.argv[1], sys.argv[2], len(a), len(b), len(same), len(diff), len(set(b) - set(a)), len(set(a) - set(b))))
for k in diff: print("  CHANGED", k[:140])
for k in sorted(set(b) - set(a)): print("  new    ", k[:140])
for k in sorted(set(a) - set(b)): print("  gone   ", k[:140])

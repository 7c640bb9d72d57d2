#!/usr/bin/env python
"""Print the instruction mix and the memory / MFMA / wait skeleton of the MFMA loop of one kernel in a .s file
(hipcc -S --cuda-device-only).  usage: isa_loop.py file.s mangled-kernel-name-substring [--full]"""
import re, sys
from collections import Counter

def main():
    s = open(sys.argv[1]).read()
    key = sys.argv[2]
    m = re.search(r'^(\S*%s\S*):' % re.escape(key), s, re.M)
    i = m.start(); j = s.index('.Lfunc_end', i)
    body = s[i:j].split('\n')
    labels = {}
    for n, l in enumerate(body):
        mm = re.match(r'^(\.LBB\d+_\d+):', l)
        if mm: labels[mm.group(1)] = n
    loops = []
    for n, l in enumerate(body):
        mm = re.search(r's_c?branch\w*\s+(\.LBB\d+_\d+)', l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < n:
            a = labels[mm.group(1)]
            if any('v_mfma' in x for x in body[a:n]): loops.append((a, n))
    a, b = (max if "--outer" in sys.argv else min)(loops, key=lambda x: x[1] - x[0])
    ins = [l.strip() for l in body[a:b + 1] if l.strip() and not l.strip().startswith(('.', ';'))]
    print(m.group(1)); print("loop: %d instructions" % len(ins)); print(Counter(x.split()[0] for x in ins).most_common(30))
    full = '--full' in sys.argv
    run = 0
    for x in ins:
        op = x.split()[0]
        if full or re.match(r'v_mfma|s_waitcnt|s_barrier|ds_|global_|buffer_|s_nop|s_cbranch', op):
            if run and not full: print("      ... %d other" % run)
            run = 0
            print("  " + x.split('//')[0].strip()[:90])
        else:
            run += 1

if __name__ == "__main__":
    main()
